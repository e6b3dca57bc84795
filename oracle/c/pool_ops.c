/* ORACLE — TEST INFRASTRUCTURE ONLY (see pool_ops.inc header).
 * Instantiates the pooling restatements for float and double. */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SFX _f32
#define CEIL_ ceilf
#define FLOOR_ floorf
#define ROUND_ roundf
/* The reference calls unqualified cos()/sin() on a float (ROIAlignRotated_cpu.cpp:232-234),
 * which resolves to the double overload and rounds the result to float; checked bit-exact
 * against the compiled reference (tests/test_oracle_vs_reference.py). */
#define COS_(x) ((float)cos((double)(x)))
#define SIN_(x) ((float)sin((double)(x)))
#include "pool_ops.inc"
#undef REAL
#undef SFX
#undef CEIL_
#undef FLOOR_
#undef ROUND_
#undef COS_
#undef SIN_

#define REAL double
#define SFX _f64
#define CEIL_ ceil
#define FLOOR_ floor
#define ROUND_ round
#define COS_ cos
#define SIN_ sin
#include "pool_ops.inc"
