/*
 * jtsm_hip.h — C ABI of libjtsm_hip.so: the MI355X (gfx950) hot path of JTSM.
 *
 * Every entry point takes plain device pointers, sizes and a hipStream_t passed as
 * void* (NULL = the null stream).  No torch/ATen types cross this boundary.  The
 * library never allocates device memory: outputs and workspaces are caller-owned,
 * exactly sized as documented; inputs are never written.  All launches are
 * asynchronous on `stream`; nothing here synchronises the device (the reference's
 * forward does a cudaDeviceSynchronize, ROIAlign_cuda.cu:364 — deliberately not kept).
 *
 * Return value: 0 on success, a negative JTSM_E* code otherwise;
 * jtsm_last_error() gives the message of the calling thread's last failure.  The
 * reference reports the same conditions as C++ exceptions -> Python RuntimeError
 * (AT_ASSERTM / TORCH_CHECK, e.g. ROIAlign_cuda.cu:318-324); the Python host layer
 * (jtsm_amd/_lib.py) turns a non-zero return into RuntimeError to keep that behaviour.
 *
 * Each group cites the reference interface it replaces (paths relative to the
 * reference tree).  `layout` selects how 4-D feature tensors are stored:
 * JTSM_NCHW is what the reference FFI uses; JTSM_NHWC is the layout the MI355X path
 * keeps activations in (channels across the 64 lanes of a wavefront -> coalesced
 * 1 KiB rows).  Pooled outputs / gradients use the same layout as the feature map:
 * (M,C,PH,PW) for NCHW, (M,PH,PW,C) for NHWC.
 */
#ifndef JTSM_HIP_H_
#define JTSM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JTSM_OK 0
#define JTSM_EINVAL (-1)   /* bad argument (shape, null pointer, unsupported combination) */
#define JTSM_ELAUNCH (-2)  /* HIP runtime reported a launch / API error */
#define JTSM_ENODEV (-3)   /* no gfx950 device / code object not loadable */

#define JTSM_NCHW 0
#define JTSM_NHWC 1

const char* jtsm_last_error(void);
/* "jtsm_hip <version> gfx950" — also proves the shared object is the HIP build. */
const char* jtsm_version(void);
/* Number of visible HIP devices (<0 on error).  Does not create a context. */
int jtsm_device_count(void);

/* Launch timing for measurement harnesses: thin hipEvent wrappers, and a hook that records `event` right after
 * the next contraction kernel launched by the calling thread (before its split-K finishing pass, if any) — the
 * kernel's own duration, as rocprofv3 reports it, without a second process. */
void* jtsm_event_create(void);
int jtsm_event_record(void* event, void* stream);
int jtsm_event_elapsed_ms(void* start, void* stop, float* ms);
void jtsm_event_destroy(void* event);
void jtsm_conv_set_mid_event(void* event);
/* Split-K finishing of the bf16x3 / fp16 contractions: 1 = inside the contraction kernel (the tile's last-arriving K
 * slice folds the slabs in slice order and runs the epilogue: sc1 write-through slab stores, sc1 loads, an agent-scope
 * ticket, no device-scope fence), 0 = the separate splitk_finish pass, -1 = follow JTSM_SPLITK_FUSED (default: 0 — the in-kernel form
 * measured 45 % slower per step on MI355X: its slab traffic goes to HBM instead of staying in L2 / Infinity Cache).
 * Both give bit-identical results (same slice order, same epilogue arithmetic); the switch exists for tests / sweeps. */
void jtsm_conv_set_splitk_fused(int mode);

/* ---------------------------------------------------------------------------
 * ROIAlign — replaces detectron2/layers/csrc/ROIAlign/ROIAlign.h:7-27
 *   ROIAlign_forward(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned)
 *   ROIAlign_backward(grad, rois, spatial_scale, pooled_h, pooled_w, B, C, H, W,
 *                     sampling_ratio, aligned)
 * bound at detectron2/layers/csrc/vision.cpp:96-97, called from
 * detectron2/layers/roi_align.py:22-59.
 * rois: (M,5) [batch_idx, x0, y0, x1, y1], same dtype as input.
 * backward: grad is dense in `layout`; grad_input (B,C,H,W in `layout`) is zero-filled
 * by the call and then accumulated with float atomics.
 * ------------------------------------------------------------------------- */
int jtsm_roi_align_forward_f32(const float* input, const float* rois, float* output, int B,
                               int C, int H, int W, int M, float spatial_scale, int pooled_h,
                               int pooled_w, int sampling_ratio, int aligned, int layout,
                               void* stream);
int jtsm_roi_align_backward_f32(const float* grad, const float* rois, float* grad_input, int B,
                                int C, int H, int W, int M, float spatial_scale, int pooled_h,
                                int pooled_w, int sampling_ratio, int aligned, int layout,
                                void* stream);
int jtsm_roi_align_forward_f64(const double* input, const double* rois, double* output, int B,
                               int C, int H, int W, int M, double spatial_scale, int pooled_h,
                               int pooled_w, int sampling_ratio, int aligned, int layout,
                               void* stream);
int jtsm_roi_align_backward_f64(const double* grad, const double* rois, double* grad_input,
                                int B, int C, int H, int W, int M, double spatial_scale,
                                int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                                int layout, void* stream);

/* ---------------------------------------------------------------------------
 * ROIAlignRotated — replaces detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated.h:7-27
 * (bound at vision.cpp:99-106, called from detectron2/layers/roi_align_rotated.py:10-47).
 * rois: (M,6) [batch_idx, cx, cy, w, h, angle_degrees]; always "aligned".
 * ------------------------------------------------------------------------- */
int jtsm_roi_align_rotated_forward_f32(const float* input, const float* rois, float* output,
                                       int B, int C, int H, int W, int M, float spatial_scale,
                                       int pooled_h, int pooled_w, int sampling_ratio,
                                       int layout, void* stream);
int jtsm_roi_align_rotated_backward_f32(const float* grad, const float* rois, float* grad_input,
                                        int B, int C, int H, int W, int M, float spatial_scale,
                                        int pooled_h, int pooled_w, int sampling_ratio,
                                        int layout, void* stream);
int jtsm_roi_align_rotated_forward_f64(const double* input, const double* rois, double* output,
                                       int B, int C, int H, int W, int M, double spatial_scale,
                                       int pooled_h, int pooled_w, int sampling_ratio,
                                       int layout, void* stream);
int jtsm_roi_align_rotated_backward_f64(const double* grad, const double* rois,
                                        double* grad_input, int B, int C, int H, int W, int M,
                                        double spatial_scale, int pooled_h, int pooled_w,
                                        int sampling_ratio, int layout, void* stream);

/* The "bit-exact ROI bin indices" contract made observable: for each of the M rois write
 * grid[2m..] = {gh, gw} and, for the first `cap` samples in (ph, pw, iy, ix) order,
 * pos[(m*cap+s)*4..] (flat y*W+x of the 4 taps, -1 when the sample is out of range) and
 * w[(m*cap+s)*4..].  Uses the very device functions the pooling kernels use.
 * rotated=0: rois (M,5) with `aligned`; rotated=1: rois (M,6). */
int jtsm_roi_sample_table_f32(const float* rois, int rotated, int M, int H, int W,
                              float spatial_scale, int pooled_h, int pooled_w,
                              int sampling_ratio, int aligned, int* grid, int* pos, float* w,
                              int cap, void* stream);

/* ---------------------------------------------------------------------------
 * MOIPool — replaces projects/WSL/wsl/layers/csrc/MOIPool/MOIPool.h:7-47
 *   MOIPool_forward(input, rois, spatial_scale, pooled_h, pooled_w, oh_labels, superpixels)
 *       -> (output, argmax)
 *   MOIPool_backward(grad, rois, argmax, spatial_scale, pooled_h, pooled_w, B, C, H, W)
 * bound at projects/WSL/wsl/layers/csrc/vision.cpp:23-24, called from
 * projects/WSL/wsl/layers/moi_pool.py:10-33.
 * rois (M,5); oh_labels (M,L) int32; superpixels (B,Hs,Ws) int32 with ids in [0,L)
 * (ids outside that range never match — the reference reads out of bounds there).
 * argmax holds the flat h*W+w of the winning cell (-1 = empty bin), stored in `layout`
 * like output.  The reference's (M,H,W) int32 `mois` temporary (MOIPool_cuda.cu:394) is
 * replaced by two bit sets kept in `workspace`: one per feature cell (which superpixels
 * lie under it) and one per roi (which superpixels are labelled 1).
 * workspace: jtsm_moi_pool_workspace_bytes(...) bytes, 16-byte aligned, contents
 * undefined on entry and exit.
 * ------------------------------------------------------------------------- */
size_t jtsm_moi_pool_workspace_bytes(int B, int H, int W, int M, int L);
int jtsm_moi_pool_forward_f32(const float* input, const float* rois, const int32_t* oh_labels,
                              const int32_t* superpixels, float* output, int32_t* argmax,
                              void* workspace, int B, int C, int H, int W, int M, int L, int Hs,
                              int Ws, float spatial_scale, int pooled_h, int pooled_w,
                              int layout, void* stream);
int jtsm_moi_pool_backward_f32(const float* grad, const float* rois, const int32_t* argmax,
                               float* grad_input, int B, int C, int H, int W, int M,
                               int pooled_h, int pooled_w, int layout, void* stream);
/* ---------------------------------------------------------------------------
 * fp16 tensors at the pooling boundary.  The reference dispatches MOIPool on half too
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF, projects/WSL/wsl/layers/csrc/MOIPool/MOIPool_cuda.cu:400,415,484) and its
 * Python layers hand half tensors to the align operators (detectron2/layers/roi_align_rotated.py:79-85: up-cast,
 * compute, cast back).  input / rois / output / grad are IEEE binary16 bit patterns; every call needs a caller
 * workspace (16-byte aligned) in which the values are widened to fp32, pooled by the fp32 kernels above and rounded
 * to fp16 once.  MOIPool is exact (a maximum of fp16 values; roi corners = round(Half(x) * Half(scale)), the product
 * rounded to half as c10::Half arithmetic does); the align operators accumulate in fp32 and round once.
 *   jtsm_pool_f16_workspace_bytes(in, rois, out, extra): elements of the tensor read, of the rois, of the tensor
 *   written (+ extra bytes) — what the align calls need (extra = 0) and the MOIPool backward needs. */
size_t jtsm_pool_f16_workspace_bytes(long in_elems, long roi_elems, long out_elems, size_t extra);
size_t jtsm_moi_pool_f16_workspace_bytes(int B, int C, int H, int W, int M, int L, int pooled_h, int pooled_w);
int jtsm_roi_align_forward_f16(const uint16_t* input, const uint16_t* rois, uint16_t* output, int B, int C, int H, int W,
                               int M, float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                               int layout, void* workspace, size_t workspace_bytes, void* stream);
int jtsm_roi_align_backward_f16(const uint16_t* grad, const uint16_t* rois, uint16_t* grad_input, int B, int C, int H,
                                int W, int M, float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                int aligned, int layout, void* workspace, size_t workspace_bytes, void* stream);
int jtsm_roi_align_rotated_forward_f16(const uint16_t* input, const uint16_t* rois, uint16_t* output, int B, int C,
                                       int H, int W, int M, float spatial_scale, int pooled_h, int pooled_w,
                                       int sampling_ratio, int layout, void* workspace, size_t workspace_bytes,
                                       void* stream);
int jtsm_roi_align_rotated_backward_f16(const uint16_t* grad, const uint16_t* rois, uint16_t* grad_input, int B, int C,
                                        int H, int W, int M, float spatial_scale, int pooled_h, int pooled_w,
                                        int sampling_ratio, int layout, void* workspace, size_t workspace_bytes,
                                        void* stream);
int jtsm_moi_pool_forward_f16(const uint16_t* input, const uint16_t* rois, const int32_t* oh_labels,
                              const int32_t* superpixels, uint16_t* output, int32_t* argmax, void* workspace,
                              size_t workspace_bytes, int B, int C, int H, int W, int M, int L, int Hs, int Ws,
                              float spatial_scale, int pooled_h, int pooled_w, int layout, void* stream);
int jtsm_moi_pool_backward_f16(const uint16_t* grad, const uint16_t* rois, const int32_t* argmax, uint16_t* grad_input,
                               void* workspace, size_t workspace_bytes, int B, int C, int H, int W, int M, int pooled_h,
                               int pooled_w, int layout, void* stream);

/* All FPN levels in ONE launch — what detectron2/modeling/poolers.py:193-250 does level by level around
 * wsl/layers/moi_pool.py:10-33.  inputs[l] / grad_inputs[l]: (B,H[l],W[l],C) NHWC maps (host arrays of device
 * pointers, nlevels <= 8); roi_level[n] in [0,nlevels) picks roi n's map; output / argmax as above (NHWC).
 * Backward: with `scales` (the forward's level scales), a workspace
 * (jtsm_moi_pool_backward_levels_workspace_bytes) and C a multiple of 256, every gradient map is produced by a
 * gather — one workgroup per 8x8-cell tile sums, in roi / bin order, the gradients of the bins whose argmax fell
 * into it (LDS accumulation, no atomics, bitwise reproducible, every cell written once) — unless a census of the
 * roi boxes estimates more than 4000 (roi, bin) pairs on one tile (proposals piled on one spot), in which case, and otherwise (scales or workspace NULL)
 * each grad_inputs[l] is zero-filled and accumulated with float atomics. */
size_t jtsm_moi_pool_levels_workspace_bytes(int B, const int* H, const int* W, int nlevels, int M, int L);
int jtsm_moi_pool_forward_levels_f32(const float* const* inputs, const int* H, const int* W,
                                     const float* scales, int nlevels, const float* rois,
                                     const int32_t* roi_level, const int32_t* oh_labels,
                                     const int32_t* superpixels, float* output, int32_t* argmax,
                                     void* workspace, int B, int C, int M, int L, int Hs, int Ws,
                                     int pooled_h, int pooled_w, void* stream);
size_t jtsm_moi_pool_backward_levels_workspace_bytes(const int* H, const int* W, int nlevels, int B, int M);
int jtsm_moi_pool_backward_levels_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                      const int32_t* argmax, float* const* grad_inputs, const int* H,
                                      const int* W, const float* scales, int nlevels, int B, int C, int M,
                                      int pooled_h, int pooled_w, int accumulate, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* mois (M,H,W) int32 exactly as MoIForward (MOIPool_cuda.cu:138-215) would write it;
 * test/diagnostic entry, same workspace contract as the forward. */
int jtsm_moi_mask_f32(const float* rois, const int32_t* oh_labels, const int32_t* superpixels,
                      int32_t* mois, void* workspace, int B, int H, int W, int M, int L, int Hs,
                      int Ws, float spatial_scale, void* stream);


/* ---------------------------------------------------------------------------
 * Convolution / linear layers on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 * The reference has no source for these: it calls ATen/cuDNN through F.conv2d /
 * nn.Linear at detectron2/layers/wrappers.py:76-78 (Conv2d.forward),
 * detectron2/modeling/backbone/resnet.py:195-211,355-359, fpn.py:127-152,
 * projects/WSL/wsl/modeling/roi_heads/box_head.py:90-93, mask_head.py:339-343,
 * detectron2/modeling/meta_arch/semantic_seg.py:170-177.  These entry points replace that
 * call and fuse what the reference runs as separate passes behind it (FrozenBatchNorm2d
 * affine, batch_norm.py:45-66; shortcut add and ReLU, resnet.py:203-210).
 *
 * Layouts: activations NHWC (batch, h, w, c) dense; weights OHWI = [out_c][kh][kw][in_c]
 * dense (a torch (O,I,kh,kw) weight stored channels_last is exactly this).  nn.Linear's
 * [out][in] weight is the 1x1 case with in_h = in_w = 1, batch = rows.
 * in_c must be a multiple of 4 (pad RGB to 4 channels); backward also needs out_c % 4 == 0.
 * All pointers 16-byte aligned.
 * ------------------------------------------------------------------------- */
typedef struct jtsm_conv_shape {
  int batch, in_h, in_w, in_c; /* input  (batch, in_h, in_w, in_c)            */
  int out_c;                   /* output (batch, out_h, out_w, out_c)          */
  int kernel_h, kernel_w;      /* out_h = (in_h + 2*pad - dilation*(kernel_h-1) - 1)/stride + 1 */
  int stride, pad, dilation;
} jtsm_conv_shape;

int jtsm_conv_out_size(const jtsm_conv_shape* s, int* out_h, int* out_w);
/* Bytes of scratch the forward (backward_data = 0) or backward-data (= 1) call of this shape can use
 * for deterministic split-K (layers with too few output tiles to fill 256 CUs are split along K into
 * per-slice slabs that a second kernel folds in slice order and finishes with the fused epilogue).
 * 0 = not split.  Passing a NULL / smaller workspace is allowed: the call then runs unsplit. */
size_t jtsm_conv_workspace_bytes(const jtsm_conv_shape* s, int backward_data);
/* Introspection for benchmarks/profiles: which kernel a call of this shape launches (kernel 0 =
 * register-staged igemm_kernel, 1 = double-buffered direct-to-LDS igemm_dma_kernel<..,2>, 2 = its
 * single-buffered form igemm_dma_kernel<..,1> used for short K sweeps), its tile and K-split.
 * role: 0 forward, 1 backward-data, 2 backward-weight. */
int jtsm_conv_plan(const jtsm_conv_shape* s, int role, int has_kscale, int* kernel, int* tile_m,
                   int* tile_n, int* splits);

/* y = relu?( conv(x, w) * scale[c] + bias[c] + residual ).  scale, bias, residual may be NULL;
 * residual has y's shape and may alias y. */
int jtsm_conv2d_forward_f32(const float* x, const float* w, float* y, const jtsm_conv_shape* s,
                            const float* scale, const float* bias, const float* residual,
                            int relu, void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv_transpose(dy * kscale[out_c], w) (+ accumulate), then zeroed where
 * relu_mask <= 0.  kscale (per out_c; the FrozenBN scale of this conv), accumulate and
 * relu_mask (both dx-shaped; accumulate may alias dx) may be NULL. */
int jtsm_conv2d_backward_data_f32(const float* dy, const float* w, float* dx,
                                  const jtsm_conv_shape* s, const float* kscale,
                                  const float* accumulate, const float* relu_mask, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* dw[o][kh][kw][i] (+)= row_scale[o] * sum_pixels dy[p][o] * x[pix(p,kh,kw)][i].
 * zero_dw != 0 clears dw first; otherwise the result is added to what dw holds.  Split over
 * the pixel axis with float atomics: the last bits depend on arrival order. */
int jtsm_conv2d_backward_weight_f32(const float* dy, const float* x, float* dw,
                                    const jtsm_conv_shape* s, const float* row_scale, int zero_dw,
                                    void* stream);

/* Split-bf16 ("bf16x3") contractions: the same fp32-in / fp32-out convolution as above (same shapes,
 * epilogue, workspace and split-K rules), computed on the bf16 matrix cores from operands that were split
 * once into two bf16 planes, hi = bf16(x), lo = bf16(x - hi), as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with
 * fp32 accumulation (relative error per product ~2^-16; the reference is plain fp32 cuDNN/ATen conv,
 * detectron2/layers/wrappers.py:62-83).  Planes are raw bf16 bit patterns (uint16_t), 16-byte aligned.
 *   jtsm_split_bf16_f32             hi/lo[i] <- src[i], same layout (activations NHWC, weights OHWI)
 *   jtsm_split_bf16_transposed_f32  w [out_c][taps][in_c] -> planes of [in_c][taps][out_c] (what the
 *                                   data gradient contracts against); row_scale (nullable, [out_c])
 *                                   multiplies row o first — the FrozenBN scale of the layer
 *   jtsm_split_bf16_paired_f32      src [rows][k] (k % 32 == 0) -> PAIRED planes: per row, blocks of [32 k of hi]
 *                                   [32 k of lo], i.e. element (row, j) has hi at row*2k + (j/32)*64 + j%32 and lo 32
 *                                   elements behind it; `planes` holds 2*rows*k elements.  One K stage of a contraction
 *                                   (32 k of both planes) is then ONE whole 128-byte line per row instead of half a line
 *                                   of each of two arrays.  The WEIGHT operand (w_hi, w_lo / wt_hi, wt_lo) of every
 *                                   bf16x3 forward / backward-data entry point may be given in this layout: it is
 *                                   recognised by w_lo == w_hi + 32 elements (separate planes are at least a whole plane
 *                                   apart; keep them more than 32 elements apart).  jtsm_split_bf16_transposed_f32 and the
 *                                   records of jtsm_split_bf16_multi_f32 write paired planes when given such a pair
 *                                   (taps * out_c % 32 == 0; a straight record then carries its row length k in `in_c`)
 *   jtsm_conv_bf16x3_eligible       1 when a shape can take this path in `role` (0 forward, 1 backward-
 *                                   data; 2 see below): the contracted channel count is a multiple of 32, or the kernel
 *                                   is 1x1 and it is a multiple of 8 */
int jtsm_split_bf16_f32(const float* src, uint16_t* hi, uint16_t* lo, long n, void* stream);
int jtsm_split_bf16_paired_f32(const float* src, uint16_t* planes, long rows, int k, void* stream);
int jtsm_split_bf16_transposed_f32(const float* w, const float* row_scale, uint16_t* hi, uint16_t* lo,
                                   int out_c, int taps, int in_c, void* stream);
/* The same two splits for MANY weights in one launch.  table: device array of `entries` records of eight
 * 64-bit words {src, hi, lo, row_scale (transposed only, may be 0), first_block, n, taps, in_c}:
 *   transposed == 0: n = element count; the record owns ceil(n / 2048) consecutive workgroups;
 *   transposed != 0: n = out_c;          it owns ceil(in_c/32) * ceil(out_c/32) * taps workgroups.
 * first_block is the running sum of those counts (first record 0), `blocks` their total. */
int jtsm_split_bf16_multi_f32(const void* table, int entries, long blocks, int transposed, void* stream);
int jtsm_conv_bf16x3_eligible(const jtsm_conv_shape* s, int role);
/* What a bf16x3 call of this shape launches (given the advertised workspace): the template arguments of
 * igemm_x3_kernel<role,WM,WN,TM,TN,NBUF> (role 0/1) or igemm_x3_wgrad_kernel<WM,WN,TM,TN,NBUF> (role 2) —
 * WM x WN wavefronts of TM x TN 32x32 MFMA tiles — and the number of K slices.  *nbuf == 0 reports the LDS-halo
 * 3x3 kernel igemm_x3_halo_kernel<role,TH,WM,WN,TN,HP16> (TH = 8 * WM / 2 patch rows, HP16 = 12 or 21). */
int jtsm_conv_bf16x3_plan(const jtsm_conv_shape* s, int role, int* wm, int* wn, int* tm, int* tn, int* nbuf,
                          int* splits);
/* y_hi / y_lo (both or neither; needs out_c % 4 == 0): the planes of the finished output y, written by the
 * epilogue for a following bf16x3 contraction — saves that layer's jtsm_split_bf16_f32 pass. */
int jtsm_conv2d_forward_bf16x3(const uint16_t* x_hi, const uint16_t* x_lo, const uint16_t* w_hi,
                               const uint16_t* w_lo, float* y, uint16_t* y_hi, uint16_t* y_lo,
                               const jtsm_conv_shape* s, const float* scale, const float* bias,
                               const float* residual, int relu, void* workspace, size_t workspace_bytes,
                               void* stream);
/* dx_hi / dx_lo (both or neither): planes of the finished dx (after accumulate / relu_mask) — with relu_mask =
 * the previous layer's ReLU output this IS that layer's gated gradient, ready for its own two contractions. */
int jtsm_conv2d_backward_data_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* wt_hi,
                                     const uint16_t* wt_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                     const jtsm_conv_shape* s, const float* accumulate,
                                     const float* relu_mask, void* workspace, size_t workspace_bytes,
                                     void* stream);
/* role 2 (backward-weight) is eligible when in_c and out_c are multiples of 8; dy / x planes as above.
 * Unlike the fp32 kernel this one is DETERMINISTIC: each pixel slice writes a partial tile into the
 * workspace (jtsm_conv_bf16x3_wgrad_workspace_bytes, 16-byte aligned) and a fixed-order pass adds them;
 * zero_dw == 0 adds the result to what dw holds. */
size_t jtsm_conv_bf16x3_wgrad_workspace_bytes(const jtsm_conv_shape* s);
int jtsm_conv2d_backward_weight_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* x_hi,
                                       const uint16_t* x_lo, float* dw, const jtsm_conv_shape* s,
                                       const float* row_scale, int zero_dw, void* workspace,
                                       size_t workspace_bytes, void* stream);

/* fp16 contractions (BASELINE.json configs[4]: "fp16 MFMA path" — fp16 operands, fp32 accumulate, fp32 results and
 * losses; the reference reaches fp16 through torch autocast, detectron2/engine/train_loop.py AMPTrainer).  Same
 * shapes, epilogue, workspace, eligibility and split-K rules as the bf16x3 entry points, but every operand is ONE
 * plane of IEEE binary16 bit patterns (uint16_t, 16-byte aligned) and a product is one v_mfma_f32_32x32x16_f16.
 *   jtsm_split_f16_f32              h[i] <- fp16(src[i] * 2^shift), same layout
 *   jtsm_split_f16_transposed_f32   as jtsm_split_bf16_transposed_f32, one plane
 *   jtsm_split_bf16_multi_f32       with a record's `lo` word 0 writes that record's fp16 plane into `hi`
 * Gradient planes carry a power-of-two factor 2^grad_shift (exact; keeps small gradients out of fp16's subnormal
 * range — the per-contraction equivalent of loss scaling): the caller splits dy with shift = grad_shift, the kernels
 * multiply the accumulator by 2^-grad_shift before the epilogue (dx, dw come out unscaled, fp32) and write the dx
 * plane (dx_h, nullable) with the factor applied again.  y_h (nullable) is the plain fp16 plane of y. */
int jtsm_split_f16_f32(const float* src, uint16_t* h, long n, int shift, void* stream);
int jtsm_split_f16_transposed_f32(const float* w, const float* row_scale, uint16_t* h, int out_c, int taps, int in_c,
                                  void* stream);
int jtsm_conv2d_forward_f16(const uint16_t* x_h, const uint16_t* w_h, float* y, uint16_t* y_h,
                            const jtsm_conv_shape* s, const float* scale, const float* bias,
                            const float* residual, int relu, void* workspace, size_t workspace_bytes,
                            void* stream);
int jtsm_conv2d_backward_data_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                  const jtsm_conv_shape* s, const float* accumulate, const float* relu_mask,
                                  int grad_shift, void* workspace, size_t workspace_bytes, void* stream);
int jtsm_conv2d_backward_weight_f16(const uint16_t* dy_h, const uint16_t* x_h, float* dw, const jtsm_conv_shape* s,
                                    const float* row_scale, int zero_dw, int grad_shift, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* ---- ConvTranspose2d(kernel 2, stride 2, padding 0, bias) — the mask heads' upsampler
 * (detectron2/modeling/roi_heads/mask_head.py:245-252 `deconv`; projects/WSL/wsl/modeling/roi_heads/mask_head.py:303-305;
 * ATen conv_transpose2d behind detectron2/layers/wrappers.py `ConvTranspose2d = torch.nn.ConvTranspose2d`).
 * Tensors are NHWC: x (batch, h, w, in_c), y (batch, 2h, 2w, out_c); the weight is the parameter (in_c, out_c, 2, 2)
 * in channels_last memory order [in_c][dy][dx][out_c].  in_c % 32 == 0, out_c % 32 == 0.
 *   forward        y[b,2i+dy,2j+dx,o] = relu?(bias[o] + sum_c x[b,i,j,c] W[c][dy][dx][o]) — one GEMM whose epilogue
 *                  writes straight to the four output pixels.  wt_* = planes of W^T as
 *                  jtsm_split_bf16_transposed_f32(W, null, .., out_c = in_c, taps = 1, in_c = 4*out_c) makes them;
 *                  y_hi / y_lo (nullable pair): planes of y for the next contraction.
 *   backward_data  dx[b,i,j,c] = sum_{dy,dx,o} g[b,2i+dy,2j+dx,o] W[c][dy][dx][o], kept where relu_mask > 0 (nullable:
 *                  the ReLU output x came from) — the forward role of the 2x2/stride-2 convolution whose OHWI weight is
 *                  the same memory: w_* = plain planes of W (jtsm_split_bf16_f32); dx_hi / dx_lo nullable pair;
 *                  gate_plane: the gate as a plane (see "planes only" below), nullable;
 *                  workspace: jtsm_conv_transpose2x2_workspace_bytes (may be null: no K split).
 *   backward_weight: jtsm_conv2d_backward_weight_bf16x3 with shape {batch, 2h, 2w, in_c = out_c, out_c = in_c, 2, 2,
 *                  stride 2, pad 0}, dy planes = x's, x planes = g's: dW comes out in the parameter's memory order.
 * The _f16 forms take one fp16 plane per operand; g's plane carries 2^grad_shift as in jtsm_conv2d_backward_data_f16. */
int jtsm_conv_transpose2x2_forward_bf16x3(const uint16_t* x_hi, const uint16_t* x_lo, const uint16_t* wt_hi,
                                          const uint16_t* wt_lo, float* y, uint16_t* y_hi, uint16_t* y_lo, int batch,
                                          int h, int w, int in_c, int out_c, const float* bias, int relu, void* stream);
int jtsm_conv_transpose2x2_forward_f16(const uint16_t* x_h, const uint16_t* wt_h, float* y, uint16_t* y_h, int batch,
                                       int h, int w, int in_c, int out_c, const float* bias, int relu, void* stream);
size_t jtsm_conv_transpose2x2_workspace_bytes(int batch, int h, int w, int in_c, int out_c);
int jtsm_conv_transpose2x2_backward_data_bf16x3(const uint16_t* g_hi, const uint16_t* g_lo, const uint16_t* w_hi,
                                                const uint16_t* w_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                                int batch, int h, int w, int in_c, int out_c, const float* relu_mask,
                                                const uint16_t* gate_plane, void* workspace, size_t workspace_bytes,
                                                void* stream);
int jtsm_conv_transpose2x2_backward_data_f16(const uint16_t* g_h, const uint16_t* w_h, float* dx, uint16_t* dx_h,
                                             int batch, int h, int w, int in_c, int out_c, const float* relu_mask,
                                             const uint16_t* gate_plane, int grad_shift, void* workspace,
                                             size_t workspace_bytes, void* stream);
/* ---- Chains that keep their activations as PLANES only (the mask heads' tower, layers/fused_blocks.py).
 * In every plane-arithmetic entry point the fp32 result pointer (y / dx) may be NULL when its planes (y_hi.. / dx_hi..)
 * are requested: the fp32 copy is then never written.  A ReLU gate can be read from the gated activation's hi (bf16)
 * or only (fp16) plane instead of its fp32 copy: `gate_plane` (nullable, 16-byte aligned, same layout as the result)
 * keeps the result where the 16-bit pattern is a positive number — the same gate (both formats round a positive fp32
 * to a positive value) at half the bytes.  jtsm_conv2d_backward_data_ex_* = jtsm_conv2d_backward_data_* with every
 * epilogue option: row_scale (nullable, one factor per result row — pixel or roi — applied first: the data gradient
 * of a layer whose input rows were rescaled, e.g. the per-roi factor in front of the box head,
 * projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:607-633), accumulate, relu_mask (fp32 gate) and gate_plane;
 * jtsm_channel_sum_planes = the bias gradient (jtsm_channel_sum_ws_f32) of a gradient held as
 * planes: out[c] = sum_r (hi + lo)[r][c] (lo null: the fp16 plane times 2^-shift); C % 8 == 0; workspace of
 * 1024 * C floats. */
int jtsm_conv2d_backward_data_ex_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* wt_hi,
                                        const uint16_t* wt_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                        const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                        const float* relu_mask, const uint16_t* gate_plane, void* workspace,
                                        size_t workspace_bytes, void* stream);
int jtsm_conv2d_backward_data_ex_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                     const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                     const float* relu_mask, const uint16_t* gate_plane, int grad_shift, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* fp16-ONLY activations (BASELINE configs[4]; the reference's AMP step keeps activations and their gradients in fp16,
 * detectron2/engine/train_loop.py:289-336): a chain of layers whose activations exist as their fp16 operand plane alone.
 *   forward_res16:        jtsm_conv2d_forward_f16 with the RESIDUAL read from an fp16 plane (a block input that has no
 *                         fp32 copy): y = relu?(conv * scale + bias + half(residual_h)); y may be NULL (plane only).
 *   backward_data_acc16:  jtsm_conv2d_backward_data_ex_f16 with the ACCUMULATE term read from an fp16 gradient plane
 *                         carrying 2^grad_shift (the shortcut path's gradient): dx = conv^T(dy) + accumulate_h * 2^-shift,
 *                         then row scale / gate as in the _ex form; dx may be NULL (plane only).
 * Both need out_c (in_c) % 4 == 0 and 16-byte aligned planes; not the strided 1x1 scatter. */
int jtsm_conv2d_forward_res16_f16(const uint16_t* x_h, const uint16_t* w_h, float* y, uint16_t* y_h,
                                  const jtsm_conv_shape* s, const float* scale, const float* bias,
                                  const uint16_t* residual_h, int relu, void* workspace, size_t workspace_bytes,
                                  void* stream);
int jtsm_conv2d_backward_data_acc16_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                        const jtsm_conv_shape* s, const float* row_scale, const uint16_t* accumulate_h,
                                        const uint16_t* gate_plane, int grad_shift, void* workspace,
                                        size_t workspace_bytes, void* stream);

/* The same data gradients (and the transposed convolution's), also leaving the COLUMN SUMS of the finished, gated
 * result: colsum (rows x in_c floats, rows = jtsm_conv_bf16x3_colsum_rows(s, role); role 0 for the transposed
 * convolution's conv shape, 1 for a data gradient; 0 rows: not available for this shape) holds one partial sum per row
 * tile of the launch, which the caller adds up in order (jtsm_colsum_fold_f32) — the bias gradient of the layer below
 * (ATen convolution_backward's grad_bias, detectron2/layers/wrappers.py:76-78), taken where that layer's output
 * gradient is written instead of by a pass over it (sums of the fp32 values: no plane shift in them). */
int jtsm_conv_bf16x3_colsum_rows(const jtsm_conv_shape* s, int role);
int jtsm_conv2d_backward_data_colsum_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* wt_hi,
                                            const uint16_t* wt_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                            const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                            const float* relu_mask, const uint16_t* gate_plane, float* colsum,
                                            void* workspace, size_t workspace_bytes, void* stream);
int jtsm_conv2d_backward_data_colsum_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                         const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                         const float* relu_mask, const uint16_t* gate_plane, int grad_shift, float* colsum,
                                         void* workspace, size_t workspace_bytes, void* stream);
int jtsm_conv_transpose2x2_backward_data_colsum_bf16x3(const uint16_t* g_hi, const uint16_t* g_lo, const uint16_t* w_hi,
                                                       const uint16_t* w_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                                       int batch, int h, int w, int in_c, int out_c,
                                                       const float* relu_mask, const uint16_t* gate_plane, float* colsum,
                                                       void* workspace, size_t workspace_bytes, void* stream);
int jtsm_conv_transpose2x2_backward_data_colsum_f16(const uint16_t* g_h, const uint16_t* w_h, float* dx, uint16_t* dx_h,
                                                    int batch, int h, int w, int in_c, int out_c, const float* relu_mask,
                                                    const uint16_t* gate_plane, int grad_shift, float* colsum,
                                                    void* workspace, size_t workspace_bytes, void* stream);
/* Streaming passes that end in operand planes (hi + lo bf16, or lo == NULL: one fp16 plane of v * 2^shift; all
 * pointers 16-byte aligned, element counts multiples of 8):
 *   jtsm_relu_backward_split_scaled_f32  g = y > 0 ? dy * scale : 0 (+ planes of g): the backward of ReLU followed by
 *       inverted dropout when y is the DROPOUT's output (positive exactly where the unit was kept and active;
 *       scale = 1 / (1 - p)) — torch.nn.functional.dropout + F.relu of the box head,
 *       projects/WSL/wsl/modeling/roi_heads/box_head.py DiscriminativeAdaptionNeck;
 *   jtsm_split_rowscale_f32              planes of src[r][c] * row_scale[r] (nothing else is written): the per-roi
 *       rescale in front of the box head (roi_heads_jtsm.py:607-633) folded into the plane split;
 *   jtsm_dropout_split_f32               y = keep(i) ? x / (1 - p) : 0 (y may alias x) + planes of y (y_hi nullable);
 *       keep(i) is a counter-based hash of (seed, i), so no mask is stored. */
int jtsm_relu_backward_split_scaled_f32(const float* dy, const float* y, float scale, float* g, uint16_t* g_hi,
                                        uint16_t* g_lo, long n, int shift, void* stream);
int jtsm_split_rowscale_f32(const float* src, const float* row_scale, long rows, int cols, uint16_t* hi, uint16_t* lo,
                            int shift, void* stream);
int jtsm_dropout_split_f32(const float* x, float* y, uint16_t* y_hi, uint16_t* y_lo, long n, float p,
                           unsigned long long seed, void* stream);
int jtsm_channel_sum_planes(const uint16_t* hi, const uint16_t* lo, float* out, long rows, int C, int shift,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Up to 8 such gradients of one width C in a single launch (+ one fold): hi[i] / lo[i] (lo NULL: fp16 planes) with
 * rows[i] rows each -> outs[i][C]; workspace of count * 1024 * C floats. */
int jtsm_channel_sum_planes_multi(const uint16_t* const* hi, const uint16_t* const* lo, float* const* outs,
                                  const long* rows, int count, int C, int shift, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* The weight gradient AND the bias gradient db[out_c] = sum over output pixels of dy (ATen convolution_backward's
 * third result / nn.Linear's bias gradient) in one contraction: the workgroups of the first column tile run one extra
 * matrix instruction per dy fragment against an all-ones fragment, so dy is not read a second time.  db comes from the
 * same planes as dW (hi + lo, or the fp16 plane times 2^-grad_shift).  Workspace:
 * jtsm_conv_bf16x3_wgrad_bias_workspace_bytes (the slabs of jtsm_conv_bf16x3_wgrad_workspace_bytes + the bias partials). */
size_t jtsm_conv_bf16x3_wgrad_bias_workspace_bytes(const jtsm_conv_shape* s);
int jtsm_conv2d_backward_weight_bias_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* x_hi,
                                            const uint16_t* x_lo, float* dw, float* db, const jtsm_conv_shape* s,
                                            const float* row_scale, int zero_dw, void* workspace,
                                            size_t workspace_bytes, void* stream);
int jtsm_conv2d_backward_weight_bias_f16(const uint16_t* dy_h, const uint16_t* x_h, float* dw, float* db,
                                         const jtsm_conv_shape* s, const float* row_scale, int zero_dw, int grad_shift,
                                         void* workspace, size_t workspace_bytes, void* stream);
/* Up to 8 weight gradients of ONE shape in a single launch (+ one finishing launch): member i is
 * dw[i] = row_scale[i][out] * sum_pixels dy_i (x) x_i (fresh results; row_scale may be NULL, or hold NULL members).
 * For the layers a network repeats (the bottleneck blocks of a ResNet stage: same ATen convolution_backward, weight
 * output only, as jtsm_conv2d_backward_weight_bf16x3): the K (pixel) axis of the whole group is cut into as many slices
 * as ONE launch needs to fill the chip — a sixth of the slabs six separate launches write and fold.  Deterministic (slabs
 * folded in slice order); results are bit-identical for a given (shape, n), not to the single-layer entry (different
 * slicing).  Workspace: jtsm_conv_bf16x3_wgrad_group_workspace_bytes(s, n); jtsm_conv_bf16x3_wgrad_group_plan reports
 * the kernel (tile: 0 = the 3x3 LDS-halo kernel, 128, 256) and the slice count. */
size_t jtsm_conv_bf16x3_wgrad_group_workspace_bytes(const jtsm_conv_shape* s, int n);
int jtsm_conv_bf16x3_wgrad_group_plan(const jtsm_conv_shape* s, int n, int* tile, int* splits);
int jtsm_conv2d_backward_weight_group_bf16x3(int n, const uint16_t* const* dy_hi, const uint16_t* const* dy_lo,
                                             const uint16_t* const* x_hi, const uint16_t* const* x_lo,
                                             float* const* dw, const float* const* row_scale,
                                             const jtsm_conv_shape* s, void* workspace, size_t workspace_bytes,
                                             void* stream);
int jtsm_conv2d_backward_weight_group_f16(int n, const uint16_t* const* dy_h, const uint16_t* const* x_h,
                                          float* const* dw, const float* const* row_scale, const jtsm_conv_shape* s,
                                          int grad_shift, void* workspace, size_t workspace_bytes, void* stream);
/* g = dy where y > 0 else 0 (fp32), and g's fp16 plane times 2^shift (n % 8 == 0) in the same pass. */
int jtsm_relu_backward_split_f16(const float* dy, const float* y, float* g, uint16_t* g_h, long n, int shift,
                                 void* stream);

/* ---------------------------------------------------------------------------
 * Bandwidth-bound helpers (no reference source: torch elementwise ops behind
 * F.relu_ / autograd, e.g. detectron2/modeling/backbone/resnet.py:196-210).
 * ------------------------------------------------------------------------- */
/* g[i] = y[i] > 0 ? dy[i] : 0   (y = the ReLU's output). */
/* g = dy where y > 0 else 0, and the bf16 hi / lo planes of g (n % 8 == 0) in the same pass. */
int jtsm_relu_backward_split_f32(const float* dy, const float* y, float* g, uint16_t* g_hi, uint16_t* g_lo,
                                 long n, void* stream);
int jtsm_relu_backward_f32(const float* dy, const float* y, float* g, long n, void* stream);
/* out[c] = sum_r g[r*C + c]  — bias gradient of a conv / linear layer. */
int jtsm_channel_sum_f32(const float* g, float* out, long rows, int C, void* stream);
/* The same for n <= 16 small (rows[b] x width) matrices in one launch, rows added in order: out (n, width) — folds the
 * per-row-tile column sums of jtsm_conv2d_backward_data_colsum_* of several layers at once. */
int jtsm_colsum_fold_f32(const float* const* parts, const int* rows, int n, int width, float* out, void* stream);
/* The same without atomics (bitwise reproducible) and ~4x faster on large inputs: row slabs are summed into
 * `workspace` (jtsm_channel_sum_workspace_bytes, 16-byte aligned) and folded in slab order, for any C and any
 * number of rows.  Falls back to the (atomic) form above only when the workspace is missing or too small. */
size_t jtsm_channel_sum_workspace_bytes(long rows, int C);
int jtsm_channel_sum_ws_f32(const float* g, float* out, long rows, int C, void* workspace, size_t workspace_bytes,
                            void* stream);
/* SGD with momentum over many tensors in ONE launch — torch.optim.SGD as detectron2/solver/build.py:110-195
 * configures it (dampening 0, no nesterov): d = g + wd*p; buf = first_step ? d : mu*buf + d; p -= lr*buf.
 * table: device array of `entries` records of eight 64-bit words {param, grad, momentum_buffer, n, first_block,
 * lr, weight_decay, momentum} (the three floats as their bit patterns in the low 32 bits); a record owns
 * ceil(n / 1024) consecutive workgroups, first_block is the running sum, `blocks` the total. */
int jtsm_sgd_momentum_multi_f32(const void* table, int entries, long blocks, int first_step, void* stream);


/* NHWC spatial helpers, C % 4 == 0 (no reference source; torch ops at
 * detectron2/modeling/backbone/resnet.py:358 (max_pool2d 3,2,1), fpn.py:133-136
 * (interpolate nearest x2 + add), fpn.py:173-185 (max_pool2d 1,2,0)). */
int jtsm_maxpool3x3s2_forward_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* gx is zero-filled by the call; the gradient goes to the first maximum of each window. */
int jtsm_maxpool3x3s2_backward_f32(const float* x, const float* gy, float* gx, int N, int H, int W,
                                   int C, void* stream);
/* 2x2 max pooling of the WSL ResNet-v2 backbone (projects/WSL/wsl/modeling/backbone/resnet_wsl_v2.py:157-165,413):
 * stride 2 = MaxPool2d(2, 2) -> (N, H/2, W/2, C); stride 1 = ZeroPad2d((0,1,0,1)) + MaxPool2d(2, 1) -> (N,H,W,C).
 * NHWC; backward is a gather (first maximum of a window wins, as ATen). */
int jtsm_maxpool2x2_forward_f32(const float* x, float* y, int N, int H, int W, int C, int stride, void* stream);
int jtsm_maxpool2x2_backward_f32(const float* x, const float* gy, float* gx, int N, int H, int W, int C,
                                 int stride, void* stream);
/* out(N,H,W,C) = lateral(N,H,W,C) + top(N,H/2,W/2,C) repeated 2x2. */
int jtsm_upsample2_add_f32(const float* top, const float* lateral, float* out, uint16_t* out_hi, uint16_t* out_lo,
                           int N, int H, int W, int C, void* stream);   /* out_hi / out_lo (optional): bf16 planes of out */
/* out = inputs[0] + inputs[1] (+ ...), n <= 4 dense tensors of `numel` floats summed in list order (HOST array of
 * device pointers), optional bf16 planes of the sum: the level sum of SemSegFPNHead.layers
 * (detectron2/modeling/meta_arch/semantic_seg.py:178-186) in one pass. */
int jtsm_sum_tensors_f32(const float* const* inputs, int n, long numel, float* out, uint16_t* out_hi, uint16_t* out_lo,
                         void* stream);
/* out(N,Ht,Wt,C) = sum over the 2x2 blocks of g(N,2Ht,2Wt,C): backward of the x2 upsample. */
int jtsm_sum2x2_f32(const float* g, float* out, int N, int Ht, int Wt, int C, void* stream);
/* scatter=0: dst(N,Ho,Wo,C) = src(N,H,W,C)[:, ::2, ::2]; scatter=1: dst(N,H,W,C) zero-filled, then
 * dst[:, ::2, ::2] = src(N,Ho,Wo,C).  Ho = (H-1)/2+1. */
int jtsm_subsample2_f32(const float* src, float* dst, int N, int H, int W, int C, int scatter,
                        void* stream);


/* ---------------------------------------------------------------------------
 * Multiple-instance-learning losses of projects/WSL, fused forward + analytic backward.
 * No native reference: they replace PyTorch arithmetic at
 *   fast_rcnn_tsm.py:573-586 (scores = softmax(C,1) * per-image softmax(D,0)),
 *   :840-854 / :346-379 (image probabilities = clamp(sum over the image's proposals), BCE)
 *   fast_rcnn_oicr.py:282-298 (weighted CE / #valid), :350-380 (weighted L1 on the gt-class
 *   deltas / R, Box2BoxTransform weights (10,10,5,5), detectron2/modeling/box_regression.py:38-71)
 * (all under projects/WSL/wsl/modeling/roi_heads/).  Rows of image i are
 * [bag_offsets[i], bag_offsets[i+1]).  Logit matrices are row-major with leading dimension
 * `ld*` (columns of a wider fused predictor output are fine).  nc, num_cls <= 192.
 * ------------------------------------------------------------------------- */
size_t jtsm_mil_workspace_bytes(int nbags, int max_bag_rows, int nc);
/* scores (R,nc) dense; img_probs (nbags,nc) clamped to [1e-6, 1-1e-6]; loss: 1 float
 * (mean over nbags*nc when mean_loss, else sum/nbags).  labels (nbags,nc) float 0/1.
 * The workspace keeps what the backward needs and must stay untouched until then. */
int jtsm_mil_forward_f32(const float* cls_logits, const float* det_logits, int ld, int nc,
                         const int32_t* bag_offsets, int nbags, int max_bag_rows,
                         const float* labels, int mean_loss, float* scores, float* img_probs,
                         float* loss, void* workspace, void* stream);
/* d_cls/d_det (R, nc) with leading dimension ld_grad = upstream[0] * dloss/dlogits
 * (upstream NULL = 1). */
int jtsm_mil_backward_f32(const float* cls_logits, const float* det_logits, int ld, int nc,
                          const int32_t* bag_offsets, int nbags, int max_bag_rows,
                          const float* upstream, float* d_cls, float* d_det, int ld_grad,
                          const void* workspace, void* stream);

size_t jtsm_oicr_workspace_bytes(void);
/* labels (R) int32 in [-1, num_cls-1] (num_cls-1 = background, -1 = ignored); weights (R);
 * proposals / gt_boxes (R,4) xyxy.  box_deltas (R, 4*(num_cls-1)) may be NULL (no box branch).
 * losses[0] = loss_cls, losses[1] = loss_box_reg, losses[2] = #rows with weight > 1e-12. */
int jtsm_oicr_forward_f32(const float* cls_logits, int ld_cls, int num_cls, const float* box_deltas,
                          int ld_box, const int32_t* labels, const float* weights,
                          const float* proposals, const float* gt_boxes, int R, float* losses,
                          void* workspace, void* stream);
/* d_cls (R,num_cls) with leading dimension ld_dcls, d_box (R,4*(num_cls-1)) with ld_dbox (may be NULL);
 * up_cls / up_box: device scalars (NULL = 1); `losses` is the forward's output. */
int jtsm_oicr_backward_f32(const float* cls_logits, int ld_cls, int num_cls, const float* box_deltas,
                           int ld_box, const int32_t* labels, const float* weights,
                           const float* proposals, const float* gt_boxes, int R, const float* losses,
                           const float* up_cls, const float* up_box, float* d_cls, int ld_dcls,
                           float* d_box, int ld_dbox, void* stream);


/* Mask targets of rectangle pseudo ground truth (get_pgt_mask, roi_heads_jtsm.py:1928-1994, with the rectangle
 * substitution of SURVEY F8; BitMasks.crop_and_resize, detectron2/structures/masks.py:169-200):
 * out[n] (side x side, 0/1 bytes) = ROIAlign(1.0, sampling_ratio 0, aligned) of the H x W bitmask "pixel centre
 * inside rects[n] shrunk by erode" over rois[n], thresholded at 0.5.  rois, rects: (N,4) boxes.  The bitmask is
 * never materialised; the sampling arithmetic is that of jtsm_roi_align_forward_f32. */
int jtsm_rect_mask_targets_f32(const float* rois, const float* rects, uint8_t* out, int N, int side, int H, int W,
                               float erode, void* stream);
/* Mask targets from superpixel evidence (object_evidence, projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:
 * 1928-1994): target n is the union of the superpixels that row oh_row[n] of oh_labels (R, L) int32 marks, in image
 * img_of[n] of superpixels (B, H, W) int32, cropped to rois[n] at side x side as BitMasks.crop_and_resize does
 * (detectron2/structures/masks.py:169-200) and thresholded at 0.5.  oh_row[n] < 0 gives an all-zero target. */
int jtsm_sp_mask_targets_f32(const float* rois, const int32_t* oh_row, const int32_t* img_of, const int32_t* oh_labels,
                             int L, const int32_t* superpixels, uint8_t* out, int N, int side, int H, int W,
                             void* stream);
/* Targets of the mask refinery (get_pgt_mask, roi_heads_jtsm.py:1997-2022): probs (N, M, M) pasted into the H x W
 * image at rois[n] (paste_masks_in_image, detectron2/layers/mask_ops.py:74-152, >= threshold) and cropped back to
 * the same box at side x side (crop_and_resize, >= 0.5), without storing the pasted image. */
int jtsm_paste_crop_targets_f32(const float* probs, const float* rois, uint8_t* out, int N, int M, int side, int H,
                                int W, float threshold, void* stream);
/* The "top_k nearest" targets of the mask branch (roi_heads_jtsm.py:840-905): near_rows (B, Gmax, top_k) int32 = for
 * every pseudo box the top_k foreground proposals (labels != bg_label) of its image by IoU, descending, ties by row,
 * -1 padded; matched_near (R) int32 = for every foreground proposal the near target (a proposal row) with the
 * highest IoU, first maximum in (pseudo box, rank) order; -1 for the others. */
int jtsm_near_targets_f32(const float* proposals, const int32_t* bag_offsets, int B, int R, const int32_t* labels,
                          int bg_label, const float* pgt_box, const int32_t* counts, int Gmax, int top_k,
                          int32_t* near_rows, int32_t* matched_near, void* stream);

/* Label preparation (no arithmetic of the model: the index / presence glue around it, which the reference writes as
 * dozens of small tensor ops).  Per-image inputs are B <= 16 device pointers with their row counts (host arrays).
 *
 * jtsm_pooler_rois_levels_f32 — ROIPooler's two helpers as one pass (detectron2/modeling/poolers.py:22-58
 *   assign_boxes_to_levels, :68-95 convert_boxes_to_pooler_format; projects/WSL/wsl/modeling/poolers.py:24-109):
 *   rois (M,5) = [image index, x0, y0, x1, y1]; level (M) = floor(canonical_level + log2(sqrt(area) /
 *   canonical_box_size + 1e-8)) clamped to [min_level, max_level], minus min_level (NaN -> 0).  Float operations in
 *   PyTorch's order (its division by a host scalar multiplies by the rounded reciprocal): same levels, bit for bit.
 * jtsm_roi_scale_f32 — the box head's per-roi factor (projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:607-633):
 *   out[m] = bins / (nvalid[m] + 1) * (objectness[m] + 1), nvalid = bins of roi m whose MOIPool argmax (channel 0) is
 *   not -1; argmax (M, bins, C) int32 channels-last.
 * jtsm_image_labels — image-level labels (projects/WSL/wsl/modeling/roi_heads/roi_heads.py:146-161 get_image_level_gt,
 *   roi_heads_jtsm.py get_image_level_gt_stuff): oh_things (B, num_classes) 0/1 from each image's gt_classes (int64),
 *   things_cls (B, num_classes) = the present classes ascending, then the absent ones ascending; things_cnt (B) = how
 *   many are present.  With sem_seg (B x pixels labels of sem_elem_bytes 8 (int64) or 1 (uint8), clamped to 0..255):
 *   the same for the stuff labels 1 .. num_stuff - 1 (0 = things, 255 = ignore), columns label - 1, list entries
 *   column + stuff_offset.  workspace: jtsm_image_labels_workspace_bytes(B). */
int jtsm_pooler_rois_levels_f32(const float* const* boxes, const int* counts, int B, int min_level, int max_level,
                                float canonical_box_size, float canonical_level, float* rois, int32_t* level,
                                void* stream);
int jtsm_roi_scale_f32(const int32_t* argmax, int bins, int C, const float* const* objectness, const int* counts, int B,
                       float* out, void* stream);
size_t jtsm_image_labels_workspace_bytes(int B);
int jtsm_image_labels(const int64_t* const* gt_classes, const int* counts, int B, int num_classes, const void* sem_seg,
                      int sem_elem_bytes, long pixels, int num_stuff, int stuff_offset, float* oh_things,
                      int32_t* things_cls, int32_t* things_cnt, float* oh_stuff, int32_t* stuff_cls, int32_t* stuff_cnt,
                      void* workspace, void* stream);

/* The foreground proposals of the mask branch (projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:754-948:
 * `select_foreground_proposals` + the gathers that follow it): rows with labels[r] != bg_label, in row order, with
 * their box, class (int64), image, matched near target (matched nullable) and (image, box) roi; counts[b] (int64) =
 * foreground rows of image b.  The fg_* buffers hold R entries, of which the first sum(counts) are written. */
int jtsm_fg_compact(const int32_t* labels, int bg_label, const int32_t* bag_offsets, int B, int R, const float* boxes,
                    const int32_t* matched, int32_t* fg_rows, float* fg_boxes, int64_t* fg_classes, int32_t* fg_img,
                    int32_t* fg_matched, float* fg_rois, int64_t* counts, void* stream);

/* Mask loss — mask_rcnn_loss (detectron2/modeling/roi_heads/mask_head.py:31-112, used by
 * projects/WSL/wsl/modeling/roi_heads/mask_head.py): mean binary cross-entropy with logits between the
 * ground-truth-class channel of logits (N,side,side,ld) NHWC (num_classes <= ld; gt_classes NULL when
 * class-agnostic, num_classes == 1) and target (N,side,side) 0/1 bytes.  out[0] = loss.  The backward writes the
 * dense dlogits (zero off the class channel).  Deterministic (fixed-order two-stage sum). */
size_t jtsm_mask_bce_workspace_bytes(void);
int jtsm_mask_bce_forward_f32(const float* logits, int ld, int num_classes, const int64_t* gt_classes,
                              const uint8_t* target, int N, int side, float* out, void* workspace,
                              void* stream);
int jtsm_mask_bce_backward_f32(const float* logits, int ld, int num_classes, const int64_t* gt_classes,
                               const uint8_t* target, int N, int side, const float* upstream,
                               float* dlogits, void* stream);


/* ---------------------------------------------------------------------------
 * FPN-level variants (NHWC only) — what ROIPooler.forward does with nonzero + index +
 * scatter per level (detectron2/modeling/poolers.py:236-247,
 * projects/WSL/wsl/modeling/poolers.py:300-320), without the host synchronisation: every
 * level's launch walks ALL M rois and serves only those with roi_level[m] == level, reading
 * that level's feature map and writing rows m of the shared (M,PH,PW,C) output (argmax) /
 * scattering rows m of the shared gradient into that level's grad_input (zero-filled by the
 * call).  Rows of other levels are left untouched.
 * ------------------------------------------------------------------------- */
int jtsm_roi_align_forward_level_f32(const float* input, const float* rois,
                                     const int32_t* roi_level, int level, float* output, int B,
                                     int C, int H, int W, int M, float spatial_scale, int pooled_h,
                                     int pooled_w, int sampling_ratio, int aligned, void* stream);
int jtsm_roi_align_backward_level_f32(const float* grad, const float* rois,
                                      const int32_t* roi_level, int level, float* grad_input, int B,
                                      int C, int H, int W, int M, float spatial_scale, int pooled_h,
                                      int pooled_w, int sampling_ratio, int aligned, void* stream);
/* Rotated boxes (M,6) on one FPN level: ROIPooler's level loop with pooler_type "ROIAlignRotated"
 * (detectron2/modeling/poolers.py:160-165,230-249; ROIAlignRotated.h:7-27 per level).  accumulate != 0: grad_input
 * already holds a gradient and is added to (nothing is cleared). */
int jtsm_roi_align_rotated_forward_level_f32(const float* input, const float* rois, const int32_t* roi_level,
                                             int level, float* output, int B, int C, int H, int W, int M,
                                             float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                             void* stream);
int jtsm_roi_align_rotated_backward_level_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                              int level, float* grad_input, int B, int C, int H, int W, int M,
                                              float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                              int accumulate, void* stream);
/* All levels' gradient maps in ONE call (the backward of ROIPooler's level loop, detectron2/modeling/poolers.py:236-247):
 * grad_inputs[l] (nullable: that level needs no gradient) is the (B,H[l],W[l],C) NHWC map of the rois with
 * roi_level[m] == l.  The tiles of every level are workgroups of one launch.
 * accumulate != 0 (here and in jtsm_moi_pool_backward_levels_f32): the maps already hold a gradient — what autograd
 * would add afterwards, another consumer's term — and this call adds to it in place (nothing is cleared; every cell's
 * read-add-write belongs to one thread, still no atomics in the gather forms, still reproducible).
 * workspace: jtsm_roi_align_backward_levels_workspace_bytes(H, W, nlevels, B, C, M) bytes of caller-owned device memory,
 * 16-byte aligned (tile census, launch plan, per-roi reach tables) — no allocation inside the library; NULL makes the
 * call take stream-ordered scratch of its own. */
size_t jtsm_roi_align_backward_levels_workspace_bytes(const int* H, const int* W, int nlevels, int B, int C, int M);
int jtsm_roi_align_backward_levels_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                       float* const* grad_inputs, const int* H, const int* W, const float* scales,
                                       int nlevels, int B, int C, int M, int pooled_h, int pooled_w, int sampling_ratio,
                                       int aligned, int accumulate, void* workspace, size_t workspace_bytes,
                                       void* stream);
int jtsm_moi_pool_forward_level_f32(const float* input, const float* rois, const int32_t* roi_level,
                                    int level, const int32_t* oh_labels, const int32_t* superpixels,
                                    float* output, int32_t* argmax, void* workspace, int B, int C,
                                    int H, int W, int M, int L, int Hs, int Ws, float spatial_scale,
                                    int pooled_h, int pooled_w, void* stream);
int jtsm_moi_pool_backward_level_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                     int level, const int32_t* argmax, float* grad_input, int B, int C,
                                     int H, int W, int M, int pooled_h, int pooled_w, void* stream);


/* ---------------------------------------------------------------------------
 * SemSegFPNHead's non-GEMM layers, NHWC (no reference source: nn.GroupNorm(32, C) applied by
 * Conv2d.forward, detectron2/layers/wrappers.py:79-82, and nn.Upsample(scale_factor=2, "bilinear",
 * align_corners=False), detectron2/modeling/meta_arch/semantic_seg.py:126-150).
 * group_norm: x,y (N, HW, C) with C/G a multiple of 4 and C/4 dividing 256; ReLU optionally folded in
 * (relu != 0: y = max(.,0), and the backward gates dy by the recomputed sign).  mean/rstd: (N,G) outputs
 * of the forward, inputs of the backward.  Reductions are two-stage in a fixed order (deterministic).
 * ------------------------------------------------------------------------- */
size_t jtsm_group_norm_workspace_bytes(int N, long HW, int C);
int jtsm_group_norm_forward_f32(const float* x, const float* gamma, const float* beta, float* y,
                                float* mean, float* rstd, void* workspace, int N, long HW, int C, int G,
                                float eps, int relu, void* stream);
/* dx_hi / dx_lo, y_hi / y_lo below (both or neither, nullable): bf16 planes of the result for a following
 * bf16x3 contraction (see jtsm_split_bf16_f32). */
int jtsm_group_norm_backward_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, float* dx, uint16_t* dx_hi,
                                 uint16_t* dx_lo, float* dgamma, float* dbeta, void* workspace, int N, long HW,
                                 int C, int G, int relu, void* stream);
/* y (N,2H,2W,C) <- x (N,H,W,C); backward is a gather (no atomics). */
int jtsm_upsample_bilinear2x_forward_f32(const float* x, float* y, uint16_t* y_hi, uint16_t* y_lo, int N, int H,
                                         int W, int C, void* stream);
int jtsm_upsample_bilinear2x_backward_f32(const float* gy, float* gx, int N, int H, int W, int C,
                                          void* stream);


/* Fused "bilinear xS up-sampling (align_corners=False) + cross_entropy(mean, ignore_index)" of
 * SemSegFPNHead.losses (detectron2/modeling/meta_arch/semantic_seg.py:179-188): the full-resolution
 * logits are never materialised.  logits: (N,Hs,Ws,ld) NHWC, 16-byte aligned, with C <= ld <= 64 classes and
 * ld % 4 == 0 (ld = channel pitch: a 56-wide map holding 54 classes); target: (N,Hs*S,Ws*S) int64.
 * out[0] = loss, out[1] = number of non-ignored pixels.  The workspace keeps one log-sum-exp per output
 * pixel for the backward, which gathers (no atomics) and writes dlogits (N,Hs,Ws,ld), pad lanes zero. */
size_t jtsm_semseg_ce_workspace_bytes(int N, int Hs, int Ws, int S);
int jtsm_semseg_ce_forward_f32(const float* logits, int ld, int C, const int64_t* target, float* out,
                               void* workspace, int N, int Hs, int Ws, int S, long ignore_index,
                               void* stream);
int jtsm_semseg_ce_backward_f32(const float* logits, int ld, int C, const int64_t* target,
                                const float* fwd_out, const float* upstream, float* dlogits,
                                const void* workspace, int N, int Hs, int Ws, int S, long ignore_index,
                                void* stream);


/* ---------------------------------------------------------------------------
 * Pseudo-ground-truth mining and proposal labelling (no native reference; PyTorch glue at
 * projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:1167-1338 (get_pgt_top_k, top_k = 1),
 * roi_heads.py:264-370 (label_and_sample_proposals: pairwise_iou + Matcher([0.5],[0,1]), no
 * sub-sampling), fast_rcnn_oicr.py:684-783 (softmax / apply_deltas of the previous branch)).
 * Proposals of image i are rows [bag_offsets[i], bag_offsets[i+1]); image i has counts[i] <= Gmax
 * present classes, listed in classes[i*Gmax ..].
 * ------------------------------------------------------------------------- */
/* lse[r] = log sum_c exp(logits[r,c]). */
int jtsm_row_lse_f32(const float* logits, int ld, int ncls, int R, float* lse, void* stream);
/* For every (image, present class): the row with the highest score[r,class] (score = scores[r,c], or
 * exp(scores[r,c] - lse[r]) when lse != NULL, i.e. softmax of logits; lowest row wins ties).  Writes the
 * row (relative to the image), its score, the class's image-level probability as weight, and the box:
 * the proposal itself, or (deltas != NULL, (R, ld_deltas) class-major 4-tuples) the proposal decoded with
 * Box2BoxTransform(10,10,5,5).apply_deltas for that class. */
int jtsm_mine_top1_f32(const float* scores, int ld, const float* lse, const float* proposals,
                       const float* deltas, int ld_deltas, const int32_t* bag_offsets,
                       const int32_t* classes, const int32_t* counts, int B, int Gmax,
                       const float* img_probs, int nprob, int32_t* out_idx, float* out_box,
                       float* out_score, float* out_weight, void* stream);
/* Pseudo semantic target (get_pgt_sem_seg, projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:2025-2070, with
 * the rectangle substitution of SURVEY F8): out (B,H,W) int64 <- 0, then every pseudo box j < counts[b] of image b
 * paints the pixels whose centre lies in its rectangle shrunk by `erode` with classes[b,j] - class_base (1..63), in
 * ascending scores[b,j] order (ties: lower j first); finally, in list order, a class left without a single pixel is
 * painted once more.  boxes (B,G,4), classes / scores (B,G), G <= 64.  workspace: jtsm_paint_sem_seg_workspace_bytes. */
size_t jtsm_paint_sem_seg_workspace_bytes(int B);
int jtsm_paint_sem_seg(const float* boxes, const int32_t* classes, const float* scores, const int32_t* counts,
                       int B, int G, int class_base, int H, int W, float erode, int64_t* out, void* workspace,
                       void* stream);
/* The same target painted from the reference's own masks (roi_heads_jtsm.py:2038-2069: get_pgt_top_k(need_mask=True)
 * -> :1333-1334 object_evidence -> :1928-1994, superpixel branch): target j of image b is proposal row
 * bag_offsets[b] + target_idx[b,j] of oh_labels (R, L) int32, and its mask is the union of the superpixels that row
 * marks: mask_j(y,x) = oh_labels[row_j][superpixels[b,y,x]] != 0 (ids outside [0, L) belong to no mask).  Paint order,
 * values and the second pass for classes left without a pixel as above.  superpixels (B,H,W) int32 at the OUTPUT size. */
int jtsm_paint_sem_seg_evidence(const int32_t* target_idx, const int32_t* bag_offsets, const int32_t* oh_labels, int L,
                                const int32_t* superpixels, const int32_t* classes, const float* scores,
                                const int32_t* counts, int B, int G, int class_base, int H, int W, int64_t* out,
                                void* workspace, void* stream);
/* For every proposal: IoU against its image's pseudo boxes (first maximum wins), label = that box's
 * class if IoU >= iou_thresh else bg_label, plus the matched index / box / weight / (optional) score. */
int jtsm_match_label_f32(const float* proposals, const int32_t* bag_offsets, int B, int R,
                         const float* pgt_box, const int32_t* classes, const int32_t* counts,
                         const float* pgt_weight, const float* pgt_score, int Gmax, float iou_thresh,
                         int bg_label, int32_t* labels, int32_t* matched, float* gt_boxes,
                         float* gt_weights, float* gt_scores, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Inference / post-processing.  None of these entry points synchronises with the host: counts
 * that depend on the data are written to device memory.
 * ------------------------------------------------------------------------------------------------------------------ */

/* OICROutputLayers.predict_probs_K + predict_boxes_K (projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:712-783):
 * probs (R, C1) <- mean over `heads` of softmax(logits[h] (R, C1)); boxes (R, Kb*4) <- Box2BoxTransform(weights)
 * .apply_deltas(mean over heads of deltas[h] (R, Kb*4), proposals (R, 4)) (detectron2/modeling/box_regression.py:78-113).
 * `logits` / `deltas` are HOST arrays of `heads` (<= 8) device pointers; boxes == NULL skips the box part. */
int jtsm_oicr_predict_f32(const float* const* logits, const float* const* deltas, int heads, int R, int C1, int Kb,
                          const float* proposals, const float* weights /* host, 4 */, float scale_clamp, float* probs,
                          float* boxes, void* stream);

/* batched_nms (detectron2/layers/nms.py:10-31 -> torchvision.ops.boxes.batched_nms / nms, pinned 0.8.1 by
 * docker/Dockerfile:25): greedy NMS within each class.  boxes (n,4) xyxy, scores (n), idxs (n) int64 in
 * [0, num_classes) (anything else drops the element), max_per_class >= the largest class population.
 * coordinate_trick: 1 = torchvision's `boxes + idxs * (boxes.max() + 1)` fp32 offsets (the n < 40000 branch),
 * 0 = plain per-class NMS (the n >= 40000 branch), 2 = choose by the number of valid elements as nms.py does.
 * keep (n) int64 <- element indices, survivors first in descending score order (equal scores: lower index first),
 * *num_keep (device) <- number of survivors; *overflow (device, optional) <- 1 if a class exceeded max_per_class
 * (the result is then invalid).  IoU test: inter / (area_a + area_b - inter) > iou_threshold, as nms_cuda.cu. */
size_t jtsm_batched_nms_workspace_bytes(int n, int num_classes, int max_per_class);
int jtsm_batched_nms_f32(const float* boxes, const float* scores, const int64_t* idxs, int n, int num_classes,
                         int max_per_class, float iou_threshold, int coordinate_trick, int64_t* keep,
                         int32_t* num_keep, int32_t* overflow, void* workspace, size_t workspace_bytes, void* stream);

/* fast_rcnn_inference_single_image (projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:100-163) in one call:
 * rows with a non-finite box or score are dropped, boxes (R, Kb*4), Kb in {1, K}, are clipped to (img_h, img_w),
 * every (row, class < K) with scores (R, K+1) > score_thresh becomes a candidate, batched_nms (above, trick mode 2),
 * top `topk` (< 0: all).  Outputs have `cap` slots (unused slots: zeros / -1): boxes (cap,4), scores, classes,
 * rows (the proposal row of each detection) and *out_count (device). */
size_t jtsm_fast_rcnn_inference_workspace_bytes(int R, int K);
int jtsm_fast_rcnn_inference_f32(const float* boxes, const float* scores, int R, int K, int Kb, float img_h,
                                 float img_w, float score_thresh, float nms_thresh, int topk, int cap,
                                 float* out_boxes, float* out_scores, int64_t* out_classes, int64_t* out_rows,
                                 int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream);

/* mask_rcnn_inference (projects/WSL/wsl/modeling/roi_heads/mask_head.py:106-147) on the head average of
 * roi_heads_jtsm.py:949-961: out (N, M, M) <- sigmoid((sum_h logits[h][n, c_n]) / heads), c_n = classes[n] (0 when
 * C == 1).  `logits`: HOST array of device pointers to (N, C, M, M). */
int jtsm_mask_probs_f32(const float* const* logits, int heads, const int64_t* classes, int N, int C, int M,
                        float* out, void* stream);

/* paste_masks_in_image (detectron2/layers/mask_ops.py:74-145; GPU branch of _do_paste_mask :17-71): out (N, img_h,
 * img_w) <- grid_sample(masks (N, M, M), bilinear, zeros, align_corners=False) over the whole image, then
 * `>= threshold` as 0/1 (threshold >= 0) or `* 255` truncated to uint8 (threshold < 0). */
int jtsm_paste_masks_f32(const float* masks, const float* boxes, int N, int M, int img_h, int img_w, float threshold,
                         uint8_t* out, void* stream);

/* F.interpolate(x[..., :crop_h, :crop_w], mode="bilinear", align_corners=False) into planar y (N, C, out_h, out_w);
 * scale_h / scale_w are the source-index scales (1 / scale_factor, or crop / out): SemSegFPNHead's x common_stride
 * upsampling (detectron2/modeling/meta_arch/semantic_seg.py:172-176) and sem_seg_postprocess
 * (detectron2/modeling/postprocessing.py:75-100). */
int jtsm_resize_bilinear_f32(const float* x, int layout, int N, int C, int H, int W, int crop_h, int crop_w,
                             int out_h, int out_w, float scale_h, float scale_w, float* y, void* stream);
/* F.interpolate(x, size=(out_h, out_w), mode="nearest") of a planar (C, H, W) map, optionally reading the source
 * columns mirrored: the resize + un-flip of GeneralizedRCNNWithTTAAVG._reduce_pred_sem_seg
 * (projects/WSL/wsl/modeling/test_time_augmentation_avg.py:428-442; ResizeTransform.apply_segmentation on float
 * arrays, detectron2/data/transforms/transform.py:116-141). */
int jtsm_resize_nearest_f32(const float* x, int C, int H, int W, int out_h, int out_w, int flip_source, float* y,
                            void* stream);
/* out (HW) int64 <- argmax over c of planar x (C, HW); first maximum wins (sem_seg_r.argmax(dim=0), mcnn.py:352). */
int jtsm_argmax_channels_f32(const float* x, int C, long HW, int64_t* out, void* stream);

/* combine_semantic_and_instance_outputs (detectron2/modeling/meta_arch/panoptic_fpn.py:133-218).  masks (N, H, W)
 * uint8, order (N) int32 = instance indices by descending score, scores / classes (N), sem (H, W) int64 in [0, S),
 * S <= 256.  panoptic (H, W) int32 <- segment ids; seg_table (N + S, 5) int32 rows {id, isthing, category_id,
 * instance_id or -1, area (stuff: unpainted area; things: newly painted pixels)}, seg_score (N + S),
 * *num_segments (device).  max_visits: number of leading instances (in `order`) to walk — pass the count of scores
 * >= the confidence threshold when it is known, or -1 for all N (the walk itself stops at the threshold too). */
size_t jtsm_panoptic_combine_workspace_bytes(int N, int S);
int jtsm_panoptic_combine(const uint8_t* masks, const int32_t* order, const float* scores, const int64_t* classes,
                          int N, int H, int W, const int64_t* sem, int S, double overlap_threshold,
                          int stuff_area_limit, float instances_confidence_threshold, int32_t* panoptic,
                          int32_t* seg_table, float* seg_score, int32_t* num_segments, int max_visits,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Model input boundary: GeneralizedMCNNWSL.preprocess_image
 * (projects/WSL/wsl/modeling/meta_arch/mcnn.py:303-318) + ImageList.from_tensors
 * (detectron2/structures/image_list.py:71-125) in one launch: images[b] is the mapper's uint8 (C, h_b, w_b) planar
 * image on the device (`images`, `heights`, `widths`, `mean`, `stdv` are HOST arrays); out (B, Hp, Wp, C) float32
 * channels-last <- (x - mean[c]) / stdv[c] inside the image, pad_value in the bottom / right padding. */
int jtsm_preprocess_images_u8(const uint8_t* const* images, const int32_t* heights, const int32_t* widths, int B,
                              int C, const float* mean, const float* stdv, float pad_value, int Hp, int Wp,
                              float* out, void* stream);
/* The same for images that arrive as float32 planes (the reference accepts any dtype: `(x - mean) / std` promotes). */
int jtsm_preprocess_images_f32(const float* const* images, const int32_t* heights, const int32_t* widths, int B,
                               int C, const float* mean, const float* stdv, float pad_value, int Hp, int Wp,
                               float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JTSM_HIP_H_ */
