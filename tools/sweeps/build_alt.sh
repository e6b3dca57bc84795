# usage: bash tools/sweeps/build_alt.sh NAME -DFLAG=... [-D...]  -> scratch/alt/NAME.so: the library with conv_igemm.hip
# recompiled under the extra flags (other objects reused) — for same-box A/B runs through tools/sweeps/ab_lib.sh.
set -e
NAME=$1; shift
mkdir -p scratch/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc "$@" \
  -c jtsm_amd/csrc/conv_igemm.hip -o scratch/alt/$NAME.o
OBJS=$(ls jtsm_amd/lib/obj/*.o | grep -v conv_igemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/alt/$NAME.so scratch/alt/$NAME.o $OBJS
ls -la scratch/alt/$NAME.so
