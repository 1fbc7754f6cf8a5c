/* gomatching_hip.h -- C ABI of libgomatching_hip.so: the MI355X (gfx950) GoMatching inference hot path.
 *
 * Drop-in boundary.  The reference (Hxyz-123/GoMatching) has exactly one native
 * interface on this path, the pybind module `adet._C` (third_party/adet/layers/csrc/vision.cpp:52-55); its
 * forward entry is replaced 1:1 by gom_ms_deform_attn_forward below.  Everything else on the path is
 * stock torch.nn in the reference; here those ops are hand-written HIP kernels behind the remaining entry
 * points, which the Python mirror of the reference's META_ARCH / ROI_HEADS classes
 * (gomatching_amd/modeling) binds through ctypes.  INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *   - plain pointers and sizes only; every tensor pointer is DEVICE memory unless marked [host];
 *   - tensors are dense row-major fp32, channels-last (NHWC / token-major), 16-byte aligned;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous and capture-safe
 *     (no allocation, no synchronisation) unless stated;
 *   - return value: GOM_OK, GOM_ERR_INVALID_ARG for a violated precondition (the reference raises a
 *     RuntimeError from AT_ASSERTM there, ms_deform_attn_cuda.cu:28-52), GOM_ERR_UNSUPPORTED, or
 *     GOM_ERR_HIP_BASE + hipError_t for a launch failure (the reference only printf()s those,
 *     ms_deform_im2col_cuda.cuh:948-952).
 */
#ifndef GOMATCHING_HIP_H_
#define GOMATCHING_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOM_OK 0
#define GOM_ERR_INVALID_ARG 1
#define GOM_ERR_UNSUPPORTED 2
#define GOM_ERR_HIP_BASE 1000

#define GOM_DTYPE_F32 0
#define GOM_DTYPE_F64 1

#define GOM_ABI_VERSION 1
int gom_abi_version(void);
/* gfx arch string of device 0's code object target, e.g. "gfx950" (static storage). */
const char* gom_built_for_arch(void);

/* ---- A7: multi-scale deformable attention, forward --------------------------------------------
 * Replaces at::Tensor ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc,
 * attn_weight, im2col_step)  (third_party/adet/layers/csrc/DeformAttn/ms_deform_attn.h:20-39 ->
 * ms_deform_attn_cuda.cu:20-80 -> ms_deform_im2col_cuda.cuh:237-299).
 *   value [batch, spatial_size, num_heads, channels] ; spatial_shapes [num_levels,2] (H,W) int64 ;
 *   level_start_index [num_levels] int64 ; sampling_loc [batch, num_query, heads, levels, points, 2] ;
 *   attn_weight [batch, num_query, heads, levels, points] ; output [batch, num_query, heads*channels]
 * The caller owns `output` (the reference allocates it with at::zeros; every element is written here).
 * im2col_step has no equivalent: the whole batch is one launch.  Any shape: 8 heads x 32 channels x 4 levels x 4 points
 * (every shipped config) runs the wave-per-query kernel of msda.hip, anything else the general kernel of msda_any.hip. */
int gom_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* sampling_loc, const float* attn_weight, float* output, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                               int num_point, void* stream);

/* The reference's dispatch on the value dtype (AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:64): dtype =
 * GOM_DTYPE_F32 | GOM_DTYPE_F64 is the element type of value, sampling_loc, attn_weight and output; anything else
 * -> GOM_ERR_UNSUPPORTED (the macro throws there).  Always the general kernel. */
int gom_ms_deform_attn_forward_any(int dtype, const void* value, const int64_t* spatial_shapes,
                                   const int64_t* level_start_index, const void* sampling_loc, const void* attn_weight,
                                   void* output, int batch, int spatial_size, int num_heads, int channels, int num_levels,
                                   int num_query, int num_point, void* stream);

/* Replaces std::vector<at::Tensor> ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc,
 * attn_weight, grad_output, im2col_step)  (ms_deform_attn.h:41-61 -> ms_deform_attn_cuda.cu:83-156 ->
 * ms_deform_im2col_cuda.cuh:301-921).  grad_output [batch, num_query, heads*channels]; the three gradients have the
 * shapes of value / sampling_loc / attn_weight, are owned by the caller and need NO zero fill (grad_value is cleared
 * in-stream here, the other two are written exactly once per element).  grad_value accumulates with hardware float
 * atomics, as the reference's does: its last bits depend on the order the waves arrive in. */
int gom_ms_deform_attn_backward(int dtype, const void* value, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, const void* sampling_loc, const void* attn_weight,
                                const void* grad_output, void* grad_value, void* grad_sampling_loc, void* grad_attn_weight,
                                int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                int num_point, void* stream);

/* Same op, value read in place from a wider row-major buffer (row / batch strides in floats). */
int gom_ms_deform_attn_forward_strided(const float* value, long value_batch_stride, int value_row_stride,
                                       const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       const float* sampling_loc, const float* attn_weight, float* output, int batch,
                                       int num_query, void* stream);

/* Fused MSDeformAttn core: gom_msda_prepare + gom_ms_deform_attn_forward_strided in one pass (locations and
 * weights never reach HBM).  raw/ref as for gom_msda_prepare with ref_levels = 1. */
int gom_msda_set_lane_distributed(int on);   /* [host] fused MSDA: 1 = per-sample arithmetic distributed over a head's lanes (default) */
int gom_msda_fused_forward(const float* raw, int ld_raw, const float* ref, const float* value, long value_batch_stride,
                           int value_row_stride, const int64_t* spatial_shapes, const int64_t* level_start_index,
                           float* output, int batch, int num_query, void* stream);
/* The same op for an ENCODER call (num_query = the pyramid's tokens, query q of a frame IS token q: the level-0 pixels in raster
 * order, then the coarser levels'; reference points = the tokens' own positions: deformable_transformer.py:288-300).  h0 x w0 and
 * h1 x w1 = host copies of spatial_shapes[0] and [1] (h1 = w1 = 0: no level-1 windows).  Level-0 and level-1 queries (tiles of
 * 8 x 16, one head per workgroup) are served from per-workgroup LDS windows of the value map -- the tile's projection on every level
 * + 5 pixels of halo; level-1 tiles (round 6) gather their level-0 samples from global memory instead (that window does not fit);
 * an octet group with a sample outside its window falls back to global memory -- the coarser levels' queries run on the kernel of
 * gom_msda_fused_forward.  Bit-identical to gom_msda_fused_forward.
 * [host] gom_msda_set_window(mask): bit 0 = level-0 windows, bit 1 = level-1 windows (default 3), 0 = always the gather kernel.
 * [host] gom_msda_window_count_fallbacks(ptr): diagnostic device word the window launches add their fallback octet groups to. */
int gom_msda_set_window(int mask);
int gom_msda_window_count_fallbacks(unsigned int* device_counter);
int gom_msda_fused_forward_encoder(const float* raw, int ld_raw, const float* ref, const float* value, long value_batch_stride,
                                   int value_row_stride, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                   float* output, int batch, int num_query, int h0, int w0, int h1, int w1, void* stream);

/* Sampling-location + softmax arithmetic of MSDeformAttn.forward (ms_deform_attn.py:136-145).
 * raw [Q, ld_raw]: columns [0,256) = sampling_offsets output, [256,384) = attention_weights logits;
 * ref [Q, ref_levels, 2] (ref_levels 1 = same point on every level, the unpadded case). */
int gom_msda_prepare(const float* raw, int ld_raw, const float* ref, int ref_levels, const int64_t* spatial_shapes,
                     float* sampling_loc, float* attn_weight, long num_query_total, void* stream);

/* ---- A2/A4/A6/A9/A10/A13/A14: dense contractions on fp32 MFMA -----------------------------------
 * C[M,N] = act( (A[+A2])[M,K] . W[N,K]^T * scale[N] + shift[N] + R[M,N] )
 * nn.Linear: W = weight, shift = bias, scale = NULL.  a_rows (optional) gathers rows of A (and A2).
 * K, lda, ldw multiples of 4. */
int gom_gemm_f32(const float* A, const float* A2, const int* a_rows, int lda, const float* W, int ldw,
                 const float* scale, const float* shift, const float* R, int ldr, int relu, float* C, int ldc, int M,
                 int N, int K, void* stream);

/* Same contraction for skinny problems (M <= 256; tracker / re-id head, weight-read bound): K split over
 * workgroups, partial sums reduced in slice order (deterministic) with the epilogue.  No A2. */
long gom_gemm_splitk_workspace_bytes(int M, int N, int K);
int gom_gemm_f32_splitk(const float* A, const int* a_rows, int lda, const float* W, int ldw, const float* scale,
                        const float* shift, const float* R, int ldr, int relu, float* C, int ldc, int M, int N, int K,
                        void* workspace, long workspace_bytes, void* stream);

/* Implicit-GEMM convolution, NHWC activations, OHWI weights [Cout, KH, KW, Cin], square kernel 1/3/7,
 * Cin a power of two >= 4.  Epilogue as above: FrozenBatchNorm as (scale, shift) (Detectron2
 * FrozenBatchNorm2d), conv bias as shift, bottleneck shortcut as R, ReLU.  Y [B, OH, OW, Cout]. */
int gom_conv2d_nhwc_f32(const float* X, const float* Wt, const float* scale, const float* shift, const float* R,
                        int relu, float* Y, int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride,
                        int pad, void* stream);

/* Tiny products (tracker association logits tgt . memory^T, transformer.py:92-96; at most 2^22 outputs): one wave per
 * output element on the VALU, K % 4 == 0, same epilogue as gom_gemm_f32 (no A2).  Deterministic. */
int gom_gemm_small_f32(const float* A, const int* a_rows, int lda, const float* W, int ldw, const float* scale,
                       const float* shift, const float* R, int ldr, int relu, float* C, int ldc, int M, int N, int K,
                       void* stream);

/* fp32-accurate variants on the bf16 matrix cores ("bf16x6": operands split into three bf16 planes, six MFMA
 * products of weight <= 2; error ~ one fp32 rounding per product; 2.67x the fp32 matrix rate).
 * The weight operand is pre-split once: planes_out [3][N][Kpad] bf16, Kpad a multiple of 32 (zero padded).
 * No second A operand; the residual R applies to columns < r_cols only (lets one launch serve fused heads). */
int gom_split_bf16x3(const float* W, int ldw, int N, int K, void* planes_out, int Kpad, void* stream);
int gom_gemm_f32_bf16x6(const float* A, const int* a_rows, int lda, const void* Wplanes, long w_plane_stride, int ldw,
                        const float* scale, const float* shift, const float* R, int ldr, int r_cols, int relu, float* C,
                        int ldc, int M, int N, int K, void* stream);
int gom_conv2d_nhwc_f32_bf16x6(const float* X, const void* Wplanes, long w_plane_stride, int ldw, const float* scale,
                               const float* shift, const float* R, int relu, float* Y, int B, int H, int Wd, int Cin,
                               int Cout, int KH, int KW, int stride, int pad, void* stream);

/* "f16x3": the same products on the fp16 matrix cores from TWO fp16 planes per operand and three MFMA passes
 * (a0b0 + a0b1 + a1b0; 22 significand bits, relative error ~7e-7 per product, half the MFMA work of bf16x6).
 * Weights: gom_split_f16x2 scales each row by a power of two into fp16's upper normal range before the split and returns
 * the inverse scales (`wscale`, applied to the accumulator).  Activations are split unscaled: |a| <= 65504, absolute
 * accuracy 3e-8 below ~0.25; a non-finite result sets *flag (device int, may be NULL) -- checked by the host once per
 * step.  Otherwise the arguments of the bf16x6 entry points; `splits` / workspace as gom_conv2d_nhwc_f32_bf16x6_splitk. */
int gom_split_f16x2(const float* W, int ldw, int N, int K, void* planes_out, int Kpad, float* inv_scale, void* stream);
int gom_gemm_f32_f16x3(const float* A, const int* a_rows, int lda, const void* Wplanes, long w_plane_stride, int ldw,
                       const float* wscale, const float* scale, const float* shift, const float* R, int ldr, int r_cols,
                       int relu, float* C, int ldc, int M, int N, int K, int* flag, void* stream);
/* ... with a PERIODIC residual: row m adds R[m % r_period] (r_period = 0: R[m]).  The encoder's position term
 * (src + pos) W^T = src W^T + (pos W^T) is such a table -- S rows shared by the frames of a step (57 MB instead of 457 MB
 * per layer call at 8 frames of 1000x1778). */
int gom_gemm_f32_f16x3_rp(const float* A, const int* a_rows, int lda, const void* Wplanes, long w_plane_stride, int ldw,
                          const float* wscale, const float* scale, const float* shift, const float* R, int ldr, int r_cols,
                          int r_period, int relu, float* C, int ldc, int M, int N, int K, int* flag, void* stream);
int gom_conv2d_nhwc_f32_f16x3(const float* X, const void* Wplanes, long w_plane_stride, int ldw, const float* wscale,
                              const float* scale, const float* shift, const float* R, int relu, float* Y, int B, int H,
                              int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad, void* workspace,
                              long workspace_bytes, int splits, int* flag, void* stream);
/* 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS (csrc/conv3x3_patch.hip): the bottleneck blocks'
 * conv2 (Detectron2 BottleneckBlock behind gom_lstmatcher.py:42-61).  Cin and Cout multiples of 64
 * (gom_conv3x3_patch_supported).  gom_conv3x3_patch_image: one-time fragment-linear image of the gom_split_f16x2 planes of
 * W [Cout, 3, 3, Cin] (gom_conv3x3_patch_image_bytes bytes; -1 = shape not served); wscale = the split's inverse row scales.
 * fp32-class like the implicit-GEMM kernel (k order: 64-channel chunk, tap, channel; 32-wide k-steps): not its bits. */
int gom_conv3x3_patch_supported(int Cin, int Cout);
long gom_conv3x3_patch_image_bytes(int Cin, int Cout);
int gom_conv3x3_patch_image(const void* w_planes, long w_plane_stride, int ldw, int Cin, int Cout, void* image, long image_bytes,
                            void* stream);
int gom_conv3x3_patch_f32_f16x3(const float* X, const void* image, const float* wscale, const float* scale, const float* shift,
                                int relu, float* Y, int B, int H, int Wd, int Cin, int Cout, int* flag, void* stream);
/* The ResNet stem as one launch (csrc/stem_pool.hip): conv 7x7 / stride 2 / pad 3 from the 4-channel NHWC input to 64 channels
 * (weights = the gom_split_f16x2 planes of the [64, 7*7*4] matrix, K padded to ldw) + per-channel scale / shift (folded
 * BatchNorm, may be NULL) + ReLU + max_pool2d(3, stride 2, pad 1): Y [B, PH, PW, 64], PH = ((H - 1) / 2) / 2 + 1 likewise PW
 * (detectron2 BasicStem: modeling/backbone/resnet.py as built by adet's build_resnet_backbone).  Equal bit
 * for bit to gom_conv2d_nhwc_f32_f16x3 followed by gom_maxpool3x3s2_nhwc_f32, same range contract and *flag. */
int gom_stem_conv_pool_f32(const float* X, const void* Wplanes, long w_plane_stride, int ldw, const float* wscale,
                           const float* scale, const float* shift, float* Y, int B, int H, int Wd, int* flag, void* stream);

/* Output projection + residual + LayerNorm of an attention block in ONE launch (csrc/proj_ln.hip):
 *     Y = LayerNorm(X W^T + b + R) * gamma + beta,   X, R, Y [M, 256] fp32 (row strides ldx / ldr / ldy; Y may alias R), W [256, 256]
 * = `norm1(src + out_proj(...))` of an encoder layer and the three `norm_*(tgt + out_proj(...))` of a decoder layer
 * (deformable_transformer.py:258-264, 386-422): 3 KB of HBM traffic per token instead of the 5 KB of GEMM + LayerNorm.  f16x3
 * scheme, range contract and *flag of gom_gemm_f32_f16x3 (fp32-class agreement with it, k-steps of 32).  gom_proj_ln_image:
 * one-time weight preparation from the gom_split_f16x2 planes (gom_proj_ln_image_bytes bytes; -1 = shape not served);
 * w_inv_scale = the split's inverse row scales, bias may be NULL. */
long gom_proj_ln_image_bytes(int n, int k);
int gom_proj_ln_image(const void* w_planes, long w_plane_stride, int ldw, int n, int k, void* image, long image_bytes,
                      void* stream);
int gom_proj_ln_f32(const float* X, int ldx, const void* image, const float* w_inv_scale, const float* bias, const float* R,
                    int ldr, const float* gamma, const float* beta, float eps, float* Y, int ldy, int M, int* flag,
                    void* stream);
/* R may be NULL in gom_proj_ln_f32 (Y = LayerNorm(X W^T + b): enc_output + enc_output_norm, deformable_transformer.py:171-172).
 * Dot form: out[m] = <LayerNorm(X[m] W^T + b) * gamma + beta, dot_w[256]> + dot_b -- the proposal class logit of every encoder
 * token (:171-175) with the normalised rows never stored; the winners' rows are recomputed by gom_proj_ln_f32 on the gathered
 * rows (row-independent arithmetic: the same bits). */
int gom_proj_ln_dot_f32(const float* X, int ldx, const void* image, const float* w_inv_scale, const float* bias,
                        const float* gamma, const float* beta, float eps, const float* dot_w, float dot_b, float* out, int M,
                        int* flag, void* stream);

/* Tail of a ResNet bottleneck block fused with the head of the next (csrc/bneck_fused.hip; Detectron2 BottleneckBlock as built
 * through gom_lstmatcher.py:42-61, STRIDE_IN_1X1 = False, FrozenBN folded):
 *     X  = relu(scale3 * (A W3^T) + shift3 + R)       A [M, k1] conv2's output, R / X [M, c4 = 4 k1] (X must not alias R)
 *     Y1 = relu(scale1 * (X W1^T) + shift1)           [M, mp]: the next block's conv1
 * X is written once and never read back.  Served (k1, mp): (64, 64 | 128), (128, 128 | 256); gom_bneck_image_bytes
 * returns -1 otherwise.  gom_bneck_image: one-time weight preparation from the gom_split_f16x2 planes of conv3's [c4, k1] and the
 * next conv1's [mp, c4] matrices; scale3 / shift3 = conv3's folded BatchNorm (its weight row scales are folded in here), scale1
 * must already hold conv1's folded BatchNorm scale TIMES its weight's inverse row scales.  f16x3 scheme, range contract and *flag
 * of gom_gemm_f32_f16x3; X equals that kernel's conv3 output up to the ReLU'd last bit, Y1 to summation order. */
long gom_bneck_image_bytes(int k1, int c4, int mp);
int gom_bneck_image(const void* w3_planes, long w3_plane_stride, int ld3, const float* w3_inv_scale, const float* scale3,
                    const float* shift3, const void* w1_planes, long w1_plane_stride, int ld1, int k1, int c4, int mp, void* image,
                    long image_bytes, void* stream);
int gom_bneck_f32(const float* A, int lda, const void* image, const float* R, int ldr, const float* scale1, const float* shift1,
                  float* X, int ldx, float* Y1, int ldy, int M, int k1, int c4, int mp, int* flag, void* stream);
/* The same pair for the wide shapes (res4: k1 = 256, c4 = 1024, mp = 256; the res3 -> res4 transition: 128, 512, 256) on
 * csrc/bneck2.hip: 16-pixel waves on the 16x16x32 MFMA shape, eight per workgroup (two per SIMD) sharing one weight ring.
 * gom_bneck2_image / gom_bneck2_f32: arguments as gom_bneck_image / gom_bneck_f32 (gom_bneck2_image_bytes: -1 = shape not served). */
long gom_bneck2_image_bytes(int k1, int c4, int mp);
int gom_bneck2_image(const void* w3_planes, long w3_plane_stride, int ld3, const float* w3_inv_scale, const float* scale3,
                     const float* shift3, const void* w1_planes, long w1_plane_stride, int ld1, int k1, int c4, int mp, void* image,
                     long image_bytes, void* stream);
int gom_bneck2_f32(const float* A, int lda, const void* image, const float* R, int ldr, const float* scale1, const float* shift1,
                   float* X, int ldx, float* Y1, int ldy, int M, int k1, int c4, int mp, int* flag, void* stream);

/* The two self-attention blocks of a DeepSolo composite decoder layer, each as ONE launch (csrc/dec_attn.hip), replacing
 * nn.MultiheadAttention + residual + LayerNorm of deformable_transformer.py:386-394 (inter = 0: attention over the
 * `group_tokens` <= 32 points of a query, q = k = X + P, v = X; rows of a group are consecutive) and :396-404 (inter = 1:
 * attention over the `group_tokens` <= 128 queries of one (frame, point), q = k = v = X, P = NULL; token t of group g is row
 * ((g / inner) * group_tokens + t) * inner + g % inner, inner = points per query):
 *     Y = LayerNorm(X + out_proj(MHA_8x32(...))) * gamma + beta,   X, P, Y [rows, 256] fp32; Y must not alias X.
 * f16x3 scheme, range contract and *flag of gom_gemm_f32_f16x3 (q, k, v, the inputs and the probabilities are split into two
 * fp16 planes; fp32 accumulation and softmax).  gom_dec_attn_image: one-time weight preparation from the gom_split_f16x2
 * planes of in_proj_weight [768, 256] (+ inverse row scales, in_proj_bias) and out_proj.weight [256, 256] (+ inverse row
 * scales, bias) and the LayerNorm's gamma / beta [256]. */
long gom_dec_attn_image_bytes(int d_model, int heads);
int gom_dec_attn_image(const void* in_planes, long in_plane_stride, int ld_in, const float* in_inv_scale, const float* in_bias,
                       const void* out_planes, long out_plane_stride, int ld_out, const float* out_inv_scale,
                       const float* out_bias, const float* gamma, const float* beta, int inter, void* image, long image_bytes,
                       void* stream);
int gom_dec_attn_f32(const float* X, int ldx, const float* P, int ldp, const void* image, float eps, float* Y, int ldy, int groups,
                     int group_tokens, int inner, int inter, int* flag, void* stream);
/* The inter-instance block (inter = 1) followed, in the same launch, by the cross attention's sampling_offsets | attention_weights
 * product on its output (deformable_transformer.py:396-404, then ms_deform_attn.py:117-131 on query = tgt + query_pos):
 *     raw [rows, 384] = (Y + P2) Wraw^T + braw        P2 = query_pos [rows, 256]
 * `image`: gom_dec_attn_raw_image_bytes() bytes, filled by gom_dec_attn_image (inter = 1) and then gom_dec_attn_raw_image (the
 * gom_split_f16x2 planes of the [384, 256] weight, its inverse row scales and bias).  group_tokens <= 128. */
long gom_dec_attn_raw_image_bytes(void);
int gom_dec_attn_raw_image(const void* raw_planes, long raw_plane_stride, int ld_raw, const float* raw_inv_scale,
                           const float* raw_bias, void* image, long image_bytes, void* stream);
int gom_dec_attn_raw_f32(const float* X, int ldx, const void* image, float eps, float* Y, int ldy, const float* P2, int ldp2,
                         float* raw, int ldraw, int groups, int group_tokens, int inner, int* flag, void* stream);

/* The same two blocks and the same contract on 16-token waves, two per SIMD (csrc/dec_attn2.hip, round 6): eight waves per workgroup
 * on v_mfma_f32_16x16x32_f16 sharing one weight ring, an attention group spread over a pair of waves (inter = 0) or all eight
 * (inter = 1) that exchange a head's K / V fragments through LDS.  Arguments as the gom_dec_attn_* entry of the same name; the images
 * differ (16x16x32 fragment order, the epilogue vectors in front) and are NOT interchangeable with gom_dec_attn_image's. */
long gom_dec_attn2_image_bytes(int d_model, int heads);
int gom_dec_attn2_image(const void* in_planes, long in_plane_stride, int ld_in, const float* in_inv_scale, const float* in_bias,
                        const void* out_planes, long out_plane_stride, int ld_out, const float* out_inv_scale,
                        const float* out_bias, const float* gamma, const float* beta, int inter, void* image, long image_bytes,
                        void* stream);
int gom_dec_attn2_f32(const float* X, int ldx, const float* P, int ldp, const void* image, float eps, float* Y, int ldy, int groups,
                      int group_tokens, int inner, int inter, int* flag, void* stream);
long gom_dec_attn2_raw_image_bytes(void);
int gom_dec_attn2_raw_image(const void* raw_planes, long raw_plane_stride, int ld_raw, const float* raw_inv_scale,
                            const float* raw_bias, void* image, long image_bytes, void* stream);
int gom_dec_attn2_raw_f32(const float* X, int ldx, const void* image, float eps, float* Y, int ldy, const float* P2, int ldp2,
                          float* raw, int ldraw, int groups, int group_tokens, int inner, int* flag, void* stream);

/* The inter-instance attention (deformable_transformer.py:396-404) for MORE than 128 queries per frame (GoMatching_PP_DSText.yaml:
 * 300), csrc/dec_inter.hip: in_proj + the 8 x 32 attention core, one workgroup per (group, head) with the head's K / V^T of all
 * `group_tokens` <= 352 tokens in LDS and an online softmax over the key blocks:
 *     O[token, 32 h .. 32 h + 31] = softmax(q_h k_h^T / sqrt(32)) v_h,    q | k | v = X . in_proj_weight^T + in_proj_bias
 * X, O [rows, 256] fp32 (token t of group g is row ((g / inner) * group_tokens + t) * inner + g % inner); `image` = the
 * gom_dec_attn_image of the block with inter = 1 (its in_proj stages are read).  out_proj + residual + LayerNorm follow as one
 * gom_proj_ln_f32 launch.  Same f16x3 scheme, range contract and *flag as gom_dec_attn_f32. */
int gom_dec_inter_heads_f32(const float* X, int ldx, const void* image, float* O, int ldo, int groups, int group_tokens,
                            int inner, int* flag, void* stream);

/* Row-resident K = 256 form of gom_gemm_f32_f16x3 for SHORT problems (the decoder's Q-side nn.Linear layers at
 * M = frames x queries x points rows: deformable_transformer.py:386-422,470-488), csrc/gemm_k256.hip:
 *     C[M, N] = act( (A [+ A2])[M, 256] . W[N, 256]^T + bias [+ R on columns < r_cols] ),   N, r_cols multiples of 32.
 * Same split scheme, plane-product order and epilogue arithmetic as gom_gemm_f32_f16x3: bit-identical results, same range
 * contract and *flag.  gom_gemm_k256_image: one-time weight preparation from the gom_split_f16x2 planes, their inverse row
 * scales and the bias (may be NULL) into the kernel's fragment-linear stream (gom_gemm_k256_image_bytes bytes; -1 = shape
 * not served).  col_groups: workgroups along N (0 = chosen from M: about two workgroups per CU). */
long gom_gemm_k256_image_bytes(int N, int K);
int gom_gemm_k256_image(const void* w_planes, long w_plane_stride, int ldw, const float* w_inv_scale, const float* bias,
                        int N, int K, void* image, long image_bytes, void* stream);
int gom_gemm_k256_f32(const float* A, const float* A2, int lda, const void* image, const float* R, int ldr, int r_cols,
                      int relu, float* C, int ldc, int M, int N, int K, int col_groups, int* flag, void* stream);
/* ... with a PERIODIC residual: r_period > 0 -> row m reads R[m % r_period] (the encoder's position-embedding table, one copy
 * per frame: deformable_transformer.py:235-248 with_pos_embed folded into the projection); r_period >= 32, table < 4 GB.
 * gom_gemm_k256_f32 = r_period 0.  Long problems (>= 512 row tiles) run the kernel's whole-line-store form, short ones its
 * 16-byte-store form -- the same bits; gom_gemm_k256_set_lines(0 / 1) forces one (tests, tools), -1 = by M. */
int gom_gemm_k256_rp_f32(const float* A, const float* A2, int lda, const void* image, const float* R, int ldr, int r_cols,
                         int r_period, int relu, float* C, int ldc, int M, int N, int K, int col_groups, int* flag,
                         void* stream);
void gom_gemm_k256_set_lines(int mode);
/* Periodic residual, several periods (frames), long problem: workgroups take the row tiles frame-interleaved per XCD, so that the
 * table rows of a position are fetched into an XCD's L2 once for all frames (1 = on, 0 = index order = default: the interleave
 * measured 3 % slower, the re-reads hit the Infinity Cache; same bits). */
void gom_gemm_k256_set_interleave(int on);

/* Fused FFN block of a DeepSolo transformer layer (deformable_transformer.py:250-251,266-273 encoder linear1/ReLU/linear2 +
 * residual + norm2; :352-354,368-369 decoder + norm3):
 *     Y = LayerNorm(X + relu(X W1^T + b1) W2^T + b2) * gamma + beta,   X, Y [M, d_model] fp32 (row strides ldx / ldy; Y may
 * alias X), d_model = 256, d_hidden a multiple of 32, on the f16x3 scheme of gom_gemm_f32_f16x3 (same accuracy and range
 * contract, same *flag).  The hidden activations never leave the CU (csrc/ffn_fused.hip).
 * gom_ffn_fused_image: one-time weight preparation -- the two gom_split_f16x2 plane sets (W1 [d_hidden, d_model] with its
 * inverse row scales, W2 [d_model, d_hidden]) and b1 re-ordered into the kernel's fragment-linear stream
 * (gom_ffn_fused_image_bytes bytes; -1 = shape not served).  gom_ffn_fused_ln_f32 additionally takes W2's inverse row
 * scales and b2. */
long gom_ffn_fused_image_bytes(int d_model, int d_hidden);
int gom_ffn_fused_image(const void* w1_planes, long w1_plane_stride, int ld1, const float* w1_inv_scale, const float* b1,
                        const void* w2_planes, long w2_plane_stride, int ld2, int d_model, int d_hidden, void* image,
                        long image_bytes, void* stream);
int gom_ffn_set_stream_cus(int cus); /* [host] compute units of the (CU-masked) stream the fused FFN is launched on; 0 = whole device */
int gom_ffn_set_half_tail(int on);   /* [host] 1 (default): the partly filled last round of a long launch as half-height (64-row) tiles; same bits */
int gom_ffn_fused_ln_f32(const float* X, int ldx, const void* image, const float* w2_inv_scale, const float* b2,
                         const float* gamma, const float* beta, float eps, float* Y, int ldy, int M, int d_model,
                         int d_hidden, int* flag, void* stream);

/* The same kernel as a plain two-layer perceptron, Y = [relu](relu(X W1^T + b1) W2^T + b2), X / Y [M, 256] (csrc/ffn_fused.hip,
 * PLAIN form): the decoder's ref_point_head (deformable_transformer.py:470-473) and the first two layers of the three-layer
 * coordinate / boundary heads (:484-488, detection_transformer_wobackbone.py:238-253).  `image` = gom_ffn_fused_image. */
int gom_mlp2_fused_f32(const float* X, int ldx, const void* image, const float* w2_inv_scale, const float* b2, int relu_out,
                       float* Y, int ldy, int M, int d_model, int d_hidden, int* flag, void* stream);

/* The row-local TAIL of a composite decoder layer as one launch (csrc/dec_tail.hip; deformable_transformer.py:352-354,368-369
 * FFN + norm3, :484-488 ctrl_point_coord MLP + reference refinement, :470-473 the NEXT layer's ref_point_head over the sine
 * embedding of the refined points):
 *     Y       = LayerNorm(X + relu(X W1^T + b1) W2^T + b2) * gamma + beta
 *     new_ref = sigmoid(W3 relu(Wc2 relu(Wc1 Y + bc1) + bc2) + b3 + inverse_sigmoid(ref))           [M, 2]
 *     qpos    = Wq2 relu(Wq1 point_pos_embed(new_ref) + bq1) + bq2                                   [M, 256]; qpos == NULL: skipped
 * `image` = gom_ffn_fused_image(FFN) | gom_ffn_fused_image_acc_order(ctrl_point_coord layers 1-2) [| gom_ffn_fused_image_acc_order(
 * ref_point_head)], concatenated (gom_dec_tail_image_bytes).  *_inv_scale / *_b2: the inverse row scales and bias of each block's
 * SECOND weight.  Same f16x3 accuracy / range contract and *flag as gom_ffn_fused_ln_f32. */
long gom_dec_tail_image_bytes(int d_model, int d_hidden, int with_qpos);
int gom_ffn_fused_image_acc_order(const void* w1_planes, long w1_plane_stride, int ld1, const float* w1_inv_scale, const float* b1,
                                  const void* w2_planes, long w2_plane_stride, int ld2, int d_model, int d_hidden, void* image,
                                  long image_bytes, void* stream);
int gom_dec_tail_f32(const float* X, int ldx, const void* image, int d_hidden, const float* w2_inv_scale, const float* b2,
                     const float* gamma, const float* beta, float eps, const float* c_inv_scale, const float* c_b2,
                     const float* W3, const float* b3, const float* ref, const float* dim_t128, const float* q_inv_scale,
                     const float* q_b2, float* Y, int ldy, float* new_ref, float* qpos, int ldq, int M, int* flag, void* stream);
/* ... with the cross-attention's out_proj + residual + norm_cross in front (deformable_transformer.py:406-422):
 *     tgt3 = LayerNorm(R + S Wo^T + bo) * p_gamma + p_beta     S = the rows sampled by the fused MSDA op, R = tgt in front of the block
 * then the chain above on tgt3.  `image` = gom_dec_tail_lin_image(out_proj planes) | gom_ffn_fused_image_acc_order x 3 (the FFN's
 * input now arrives in accumulator order too).  Y must not alias S or R (it holds tgt3 while the FFN runs). */
long gom_dec_tail_lin_image_bytes(void);
int gom_dec_tail_lin_image(const void* w_planes, long w_plane_stride, int ld, void* image, long image_bytes, void* stream);
int gom_dec_tail_proj_f32(const float* S, int lds, const float* R, int ldr, const void* image, int d_hidden,
                          const float* p_inv_scale, const float* p_bias, const float* p_gamma, const float* p_beta, float p_eps,
                          const float* w2_inv_scale, const float* b2, const float* gamma, const float* beta, float eps,
                          const float* c_inv_scale, const float* c_b2, const float* W3, const float* b3, const float* ref,
                          const float* dim_t128, const float* q_inv_scale, const float* q_b2, float* Y, int ldy, float* new_ref,
                          float* qpos, int ldq, int M, int* flag, void* stream);

/* The same tail as a CU-COOPERATIVE launch (csrc/dec_tail2.hip, round 6): a workgroup's four waves share 80 rows (held in LDS as
 * MFMA operand fragments) and split the output columns; every wave streams its quarter of the weights straight from L2.  At the
 * decoder's M = 20 000 rows: 250 workgroups = one round of the chip (gom_dec_tail_*: 157 workgroups, two row groups on the busiest
 * SIMD).  `image` = `waves` per-wave streams of `wave_bytes` each (gom_dec_tail2_wave_bytes), filled block by block at `offset_bytes`
 * inside every stream: [gom_dec_tail2_image_lin(out_proj)] | gom_dec_tail2_image_mlp(FFN) | ..._mlp(ctrl_point_coord layers 1-2) |
 * [..._mlp(ref_point_head)].  S / R / p_*: R == NULL = no out_proj block (S is then tgt behind norm_cross); qpos == NULL = the last
 * layer (image without ref_point_head's part).  *_inv1 / *_b1, *_inv2 / *_b2: inverse row scales (gom_split_f16x2) and biases of each
 * block's first / second weight.  Same f16x3 accuracy / range contract and *flag as gom_dec_tail_f32; the LayerNorm statistics are
 * combined from per-wave (mean, M2) pairs (Chan), so results agree with gom_dec_tail_* to rounding, not bit for bit.
 * `waves` = 4 (one wave per SIMD; a wave owns 64 output columns) or 8 (two per SIMD, 32 columns each: one wave's epilogues and
 * activation phases under the other's MFMAs); the image is `waves` streams and belongs to the `waves` it was built for. */
long gom_dec_tail2_wave_bytes(int d_model, int d_hidden, int with_proj, int with_qpos, int waves);
int gom_dec_tail2_image_lin(const void* w_planes, long w_plane_stride, int ld, void* image, long wave_bytes, long offset_bytes,
                            int waves, void* stream);
int gom_dec_tail2_image_mlp(const void* w1_planes, long w1_plane_stride, int ld1, const void* w2_planes, long w2_plane_stride, int ld2,
                            int d_hidden, void* image, long wave_bytes, long offset_bytes, int waves, void* stream);
int gom_dec_tail2_f32(const float* S, int lds, const float* R, int ldr, const void* image, long wave_bytes, int d_hidden,
                      const float* p_inv_scale, const float* p_bias, const float* p_gamma, const float* p_beta, float p_eps,
                      const float* w1_inv_scale, const float* b1, const float* w2_inv_scale, const float* b2, const float* gamma,
                      const float* beta, float eps, const float* c_inv1, const float* c_b1, const float* c_inv2, const float* c_b2,
                      const float* W3, const float* b3, const float* ref, const float* dim_t128, const float* q_inv1,
                      const float* q_b1, const float* q_inv2, const float* q_b2, float* Y, int ldy, float* new_ref, float* qpos,
                      int ldq, int M, int waves, int* flag, void* stream);

/* Split-K form for convolutions with few output tiles and a long K (input_proj[3]: 3x3 s2 2048 -> 256 on res5, M = 3584,
 * K = 18432): `splits` K-slices run as separate workgroups into workspace [splits][M][Cout] fp32, a second kernel sums
 * them in slice order (deterministic) and applies the epilogue.  gom_conv_bf16x6_splits: recommended slice count, 0 =
 * do not split. */
int gom_conv_bf16x6_splits(int M, int N, int K);
int gom_conv2d_nhwc_f32_bf16x6_splitk(const float* X, const void* Wplanes, long w_plane_stride, int ldw,
                                      const float* scale, const float* shift, const float* R, int relu, float* Y, int B,
                                      int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                      void* workspace, long workspace_bytes, int splits, void* stream);

/* ---- normalisation ---------------------------------------------------------------------------------*/
/* out = LayerNorm(x + residual) * gamma + beta over rows of dim 256 or 1024 (residual may be NULL). */
int gom_layernorm_f32(const float* x, const float* residual, const float* gamma, const float* beta, float* out,
                      long rows, int dim, float eps, void* stream);
/* GroupNorm(32, 256) over [B, HW, 256]; writes out[b*out_batch_stride + r*256 + c] (lets the caller scatter a
 * level straight into the flattened multi-level token buffer).  stats_ws: B*64 doubles of scratch. */
int gom_groupnorm32_nhwc_f32(const float* x, const float* gamma, const float* beta, double* stats_ws, float* out,
                             long out_batch_stride, int B, int HW, int channels, float eps, void* stream);

/* ---- attention core: O = softmax(Q K^T / sqrt(head_dim)) V per (batch, head) -------------------------
 * batch = batch_outer x batch_inner.  Element (bo, bi, i, h, d) of q lives at
 * q[bo*S[0] + bi*S[1] + i*S[2] + h*head_dim + d]; strides [host] S[12] = q(bo,bi,seq) k(..) v(..) o(..).
 * head_dim 32 (DeepSolo decoder, deformable_transformer.py:388-402) or 128 (matcher, transformer.py:208,287). */
int gom_mha_core_f32(const float* q, const float* k, const float* v, float* o, int batch_outer, int batch_inner,
                     int heads, int head_dim, int Lq, int Lk, const long* strides, void* stream);

/* Ragged batch of independent attention problems over row ranges of shared q / k / v / o matrices (every frame pair
 * of the short-term matcher at once).  segments [device] int32 [S][4] = (first query row, Lq, first key row, Lk);
 * ld_* row strides in floats; head_dim 128. */
int gom_mha_core_segments_f32(const float* q, const float* k, const float* v, float* o, const int* segments,
                              int num_segments, int heads, int head_dim, int ld_q, int ld_k, int ld_v, int ld_o, int max_Lq,
                              int max_Lk, void* stream);

/* ---- glue (A1, A2 stem pool, A3, A5, A8, A9) ----------------------------------------------------------*/
/* mean3/std3 are [host] arrays.  images [B,3,H,W] -> out [B,H,W,4] (4th channel 0). */
int gom_preprocess_nchw_to_nhwc4(const float* images, const float* mean3, const float* std3, float* out, int B, int H,
                                 int W, void* stream);
/* ---- f2: frame ingest (text_track_visualizer.py:315-324 + gom_lstmatcher.py:159-170) -----------------------
 * Replaces the host-side `aug.get_transform(frame).apply_image(frame)` (Detectron2 ResizeShortestEdge ->
 * Pillow Image.resize(BILINEAR) on uint8), the BGR->RGB flip (:316-318), `astype("float32")` (:321) and
 * the model's normaliser for uint8 frames resident in HBM.  Bit-exact with Pillow's fixed-point resampler.
 * Coefficient tables come from the [host] helper (Pillow Resample.c precompute_coeffs/normalize_coeffs_8bpc):
 * bounds [out,2] = (first tap, tap count), kk [out,ksize] 22-bit fixed point; upload both before the launch. */
int gom_resample_ksize_bilinear(int in_size, int out_size);                        /* [host], -1 on bad sizes */
int gom_resample_coeffs_bilinear(int in_size, int out_size, int* bounds, int* kk, int ksize);   /* [host] */
/* src [B,H,W,3] u8 -> dst [B,OH,OW,3] u8 (flip_channels != 0 swaps channels 0 and 2). */
int gom_resize_bilinear_u8_hwc3(const uint8_t* src, int B, int H, int W, const int* xbounds, const int* xkk,
                                int xksize, const int* ybounds, const int* ykk, int yksize, uint8_t* dst, int OH,
                                int OW, int flip_channels, void* stream);
/* src [B,H,W,3] u8 -> dst [B,OH,OW,4] f32 = (resized[flipped] - mean) / std, 4th channel 0 (the stem's layout). */
int gom_ingest_u8_hwc3_to_nhwc4(const uint8_t* src, int B, int H, int W, const int* xbounds, const int* xkk,
                                int xksize, const int* ybounds, const int* ykk, int yksize, const float* mean3,
                                const float* std3, float* dst, int OH, int OW, int flip_channels, void* stream);
int gom_maxpool3x3s2_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, void* stream);
/* out [H*W, 256] = PositionalEncoding2D(normalize=True) + level_embed, for an unpadded H x W level. */
int gom_pos_encoding_2d_f32(const float* dim_t128, const float* level_embed256, float* out, int H, int W,
                            void* stream);
int gom_point_pos_embed_f32(const float* pts, const float* dim_t128, float* out, long num_points, void* stream);
/* out[q,c] = sigmoid(delta[q*ld_delta + c] + inverse_sigmoid(ref[q, c%2])), C in {2,4}. */
int gom_ref_sigmoid_f32(const float* delta, int ld_delta, const float* ref, float* out, long num_points, int C,
                        void* stream);
/* Reference refinement of a decoder layer + the next layer's point position embedding in one launch
 * (deformable_transformer.py:484-488, :470-473): new_ref[q] = sigmoid(h[q] . W3^T + b3 + inverse_sigmoid(ref[q])),
 * W3 [2,256]; pos [Q,256] (may be NULL) = gom_point_pos_embed_f32 of new_ref * (sx, sy). */
int gom_ref_update_f32(const float* h, int ld_h, const float* W3, const float* b3, const float* ref,
                       const float* dim_t128, float sx, float sy, float* new_ref, float* pos, long num_points,
                       void* stream);
int gom_proposal_valid(const int64_t* spatial_shapes, const int64_t* level_start_index, int num_levels,
                       unsigned char* valid, long S, void* stream);
int gom_encoder_reference_points(const int64_t* spatial_shapes, const int64_t* level_start_index, int num_levels,
                                 float* ref, long S, void* stream);
/* coord_raw: [B,S,8] indexed by token (compact = 0) or [B*num_queries,8] rows of the selected tokens (compact = 1). */
int gom_bezier_reference_points(const float* coord_raw, const int* topk_idx, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, int num_levels, const float* bernstein, float* refs,
                                int B, long S, int num_queries, int num_points, int compact, void* stream);
/* A18: in-place (x, y) pair scaling, detector_postprocess gom_lstmatcher.py:100-109. */
int gom_scale_xy_f32(float* x, long n_pairs, float sx, float sy, void* stream);
int gom_add_f32(const float* a, const float* b, float* out, long n, void* stream);
/* *flag |= 1 when any of x[0..n) is Inf / NaN: the result check of the bf16x6 / exact-fp32 back-ends, whose GEMM kernels carry
 * no range flag (the reference lets a non-finite activation flow on silently, gom_lstmatcher.py:268-351; here it raises). */
int gom_flag_nonfinite_f32(const float* x, long n, int* flag, void* stream);
/* dst[i] = src[i] over 32-bit words, as a kernel; src may be pinned host memory (device-mapped). */
int gom_copy_words(const void* src, void* dst, long n_words, void* stream);
int gom_broadcast_rows_f32(const float* src, float* out, long n, int B, void* stream);

/* top-k token indices per batch element (deformable_transformer.py:188-190); idx_out [B,k] int32, sorted by
 * logit descending; rows_out (optional) [B,k] = b*S + idx.  valid/invalid_logit (optional): tokens with valid[s]==0 take *invalid_logit. */
long gom_topk_workspace_bytes(int B, long S, int k);
int gom_topk_tokens(const float* logits, int ld, const unsigned char* valid, const float* invalid_logit, int B, long S,
                    int k, void* workspace, int* idx_out, int* rows_out, void* stream);

/* ---- A11/A12: detection() + NMS + foreground filter ---------------------------------------------------*/
int gom_argmax_rows_f32(const float* x, int ld, int V, long rows, int* out, void* stream);
/* Per frame b: count[b] kept instances, in NMS (descending score) order, written to the first count[b] slots of
 * the nq-padded outputs: keep_idx (row into [B*nq]), scores, boxes [.,4] px, ctrl_out [.,P,2] px,
 * bd_out [.,P,4] px, recs_out [.,P] int64. */
int gom_detect_post(const float* cls_logits, int ld_cls, const float* rescoring_logits, int ld_rescoring,
                    const float* ctrl_points, const float* bd_points, const int* recs, int B, int num_queries,
                    int num_points, float img_h, float img_w, float det_thresh, float nms_thresh, float asso_thresh,
                    int* count, int* keep_idx, float* scores, float* boxes, float* ctrl_out, float* bd_out,
                    long long* recs_out, void* stream);

/* ---- padded batches (gom_lstmatcher.py:63-76; deformable_transformer.py:141-148): every level's valid region is a
 * top-left rectangle valid_shapes [L][2] = (Hv, Wv) of its (H, W).  Mask-aware forms of the geometry tables, the fused
 * MSDA (reference points scaled by the level's valid ratio [L][2] = (Wv/W, Hv/H)) and the value zero-fill. */
int gom_pos_encoding_2d_valid_f32(const float* dim_t128, const float* level_embed256, float* out, int H, int W, int valid_h,
                                  int valid_w, void* stream);
int gom_proposal_valid_masked(const int64_t* spatial_shapes, const int64_t* level_start_index, int num_levels,
                              const int64_t* valid_shapes, unsigned char* valid, long S, void* stream);
int gom_encoder_reference_points_masked(const int64_t* spatial_shapes, const int64_t* level_start_index,
                                        const int64_t* valid_shapes, int num_levels, float* ref, long S, void* stream);
int gom_bezier_reference_points_masked(const float* coord_raw, const int* topk_idx, const int64_t* spatial_shapes,
                                       const int64_t* level_start_index, const int64_t* valid_shapes, int num_levels,
                                       const float* bernstein, float* refs, int B, long S, int num_queries, int num_points,
                                       int compact, void* stream);
int gom_msda_fused_forward_vr(const float* raw, int ld_raw, const float* ref, const float* value, long value_batch_stride,
                              int value_row_stride, const int64_t* spatial_shapes, const int64_t* level_start_index,
                              const float* valid_ratios, float* output, int batch, int num_query, void* stream);
/* zero columns [col0, col0+ncols) of the rows of buf [B*S, ld] whose token lies outside its level's valid region
 * (value.masked_fill(padding_mask, 0), ms_deform_attn.py:134-135). */
int gom_zero_padded_tokens_f32(float* buf, int ld, int col0, int ncols, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const int64_t* valid_shapes, int num_levels, int B, long S,
                               void* stream);

/* ---- f3: Swin-T backbone glue (third_party/adet/modeling/swin/swin_transformer.py) ---------------------------------
 * The linear layers use the GEMM entry points above.  Tokens are [B,H,W,C] channels-last.  Window rows are ordered
 * (image, window row, window column, token 0..48), windows of 7x7 over the map zero-padded to multiples of 7. */
int gom_layernorm_any_f32(const float* x, const float* gamma, const float* beta, float* out, long rows, int dim, float eps,
                          void* stream);                                                /* dim % 4 == 0, <= 2048 */
int gom_gelu_f32(float* x, long n, void* stream);                                         /* in place, exact erf form */
/* PatchEmbed input (:473-479): image [B,H,W,4] -> rows [B*ceil(H/4)*ceil(W/4), 64] in (kh, kw, c) order, zero padded. */
int gom_swin_patchify_f32(const float* img, float* out, int B, int H, int W, void* stream);
/* pad + cyclic shift (-shift) + window_partition (:246-262) and its inverse fused with the residual add (:264-279). */
int gom_swin_window_gather_f32(const float* x, float* out, int B, int H, int W, int C, int shift, void* stream);
int gom_swin_window_scatter_add_f32(const float* windows, const float* shortcut, float* out, int B, int H, int W, int C,
                                    int shift, void* stream);
/* PatchMerging gather (:320-327): [B,H,W,C] -> [B,ceil(H/2),ceil(W/2),4C]. */
int gom_swin_patch_merge_f32(const float* x, float* out, int B, int H, int W, int C, void* stream);
/* WindowAttention core (:139-165): qkv [num_windows*49, 3C] -> out [num_windows*49, C]; bias [heads,49,49]; mask
 * [windows_per_image,49,49] (0 / -100) or NULL; head_dim 32. */
int gom_swin_window_attention_f32(const float* qkv, float* out, const float* bias, const float* mask, long num_windows,
                                  int windows_per_image, int heads, int C, void* stream);

/* ---- f3: ViTAEv2-S backbone glue (third_party/adet/modeling/vitae_v2/) ----------------------------------------------
 * Linear layers, dense convolutions and the two products of the full attention use the GEMM entry points above. */
/* x [B,H,W,C] -> rows [B*OH*OW, ldo], columns (kh, kw, c) then zeros up to ldo: feeds the dilated strided convolutions
 * of the pyramid reduction module (PRM, ReductionCell.py:27-34,55-62) to a GEMM.  C % 4 == 0, ldo % 4 == 0. */
int gom_im2col_nhwc_f32(const float* x, float* out, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                        int dilation, int ldo, void* stream);
/* grouped 3x3 convolution, padding 1 (PCM, ReductionCell.py:97-105 / NormalCell.py:137-145): w [3,3,Cout,Cin/groups]
 * with Cin/groups in {4,16}; y = act(conv*scale + shift) + R; scale / R may be NULL; act 0 none, 3 SiLU. */
int gom_grouped_conv3x3_nhwc_f32(const float* x, const float* w, const float* scale, const float* shift, const float* R,
                                 int act, float* y, int B, int H, int W, int Cin, int Cout, int groups, int stride,
                                 void* stream);
int gom_silu_f32(float* x, long n, void* stream);                                         /* in place, n % 4 == 0 */
/* centred zero padding to multiples of 7 + window_partition, and window_reverse + crop (+ up to two addends of the token
 * shape, NULL to skip) (ReductionCell.py:147-163, NormalCell.py:160-211). */
int gom_vitae_window_gather_f32(const float* x, float* out, int B, int H, int W, int C, void* stream);
int gom_vitae_window_crop_f32(const float* windows, const float* R1, const float* R2, float* out, int B, int H, int W, int C,
                              void* stream);
/* WindowAttention core without position bias / mask (window.py:92-124): qkv [num_windows*49, 3C] -> out
 * [num_windows*49, C]; head_dim C/heads in {64,128}. */
int gom_vitae_window_attention_f32(const float* qkv, float* out, long num_windows, int heads, int C, void* stream);
/* in place x[r, :cols] = softmax(x[r, :cols] * scale), cols <= 8192 (full attention, NormalCell.py:52-54). */
int gom_softmax_rows_scaled_f32(float* x, long rows, int cols, long ld, float scale, void* stream);
/* out[c*ldo + r] = x[r*ld + c]. */
int gom_transpose_f32(const float* x, float* out, int rows, int cols, long ld, long ldo, void* stream);
/* Full self-attention with the scores kept on the CU (NormalCell.py:46-58 / token_transformer.py:27-44):
 * out[b*N + i, h*hd : (h+1)*hd] = softmax_j(q_i . k_j / sqrt(hd)) v_j over the N tokens of image b, for every head h.
 * q, k, v point at column 0 of their first head inside row-strided fp32 buffers (row stride ld, e.g. a fused qkv buffer
 * with q = qkv, k = qkv + C, v = qkv + 2C); hd in {64, 128}; products on the fp16 matrix cores with the f16x3 split
 * (operands within +-65504; a non-finite output sets *flag, which may be NULL). */
int gom_flash_attention_f32(const float* q, const float* k, const float* v, float* out, int batch, int N, int heads,
                            int head_dim, int ld, int ldo, int* flag, void* stream);

/* ---- A14/A15: tracker ---------------------------------------------------------------------------------*/
int gom_gather_rows_f32(const float* src, const int* rows, float* out, int n, int dim, void* stream);
/* per-frame softmax with an appended zero logit (lstmatcher.py:373-381); frame_offsets [num_frames+1] int32. */
int gom_asso_activate_f32(const float* logits, int ld, const int* frame_offsets, int num_frames, int n_k, float* out,
                          int ld_out, void* stream);
/* traj[n_k, M] (gom_lstmatcher.py:429-445 / 510-547).  meta int32: nonk[Np] | col_of[Np] | last_idx[M] | k_inds[n_k];
 * boxes [N,4] in pixels of the network input (img_w x img_h). */
int gom_track_score_f32(const float* act, int ld, const int* meta, const float* decay, const float* boxes, float img_w,
                        float img_h, int n_k, int Np, int M, int with_iou, float max_center_dist, float* traj,
                        void* stream);
/* gom_asso_activate_f32 + gom_track_score_f32 of one match as ONE launch (same values; Np + n_k <= 16 384). */
int gom_asso_score_f32(const float* logits, int ld, const int* frame_offsets, int num_frames, const int* meta,
                       const float* decay, const float* boxes, float img_w, float img_h, int n_k, int Np, int M,
                       int with_iou, float max_center_dist, float* traj, void* stream);
/* Short-term matching for ALL (previous, current) frame pairs of a clip in one launch: per current detection i of pair
 * p, q.k^T logits against the previous frame's rows, softmax with the zero background logit (lstmatcher.py:373-381) and
 * S[i,j] = max(a_j, IoU(i,j)) (gom_lstmatcher.py:429-445 for tracks seen once).  pairs [device] int32 [P][6] =
 * (first memory row, n_prev, n_cur, first tgt row, first box row, offset of the pair's [n_cur, n_prev] block in S);
 * row_pair [total_cur_rows]; boxes [rows,4] px in memory-row order; max_prev <= 320. */
int gom_short_term_pairs_f32(const float* tgt, const float* memory, int d, const int* pairs, const int* row_pair,
                             const float* boxes, float img_w, float img_h, int with_iou, int total_cur_rows, int max_prev,
                             float* S, void* stream);
/* [host runtime] The whole device chain of one association match -- gather of the window's embeddings, the matcher
 * transformer (roi_heads/transformer.py:60-96: n_enc post-norm encoder layers over all N rows, n_dec cross-attention
 * decoder layers for the query rows [lo, hi), norms Identity), q.k^T logits, `_activate_asso` (lstmatcher.py:373-381)
 * and the trajectory score (gom_lstmatcher.py:429-445 / 510-547) -- queued on `stream` by ONE call.  Replaces ~18
 * per-kernel FFI crossings of the launch-bound tracker recurrence.  Layer weights are nn.MultiheadAttention / Linear
 * tensors as stored in the checkpoint (in_w [3d,d], out_w [d,d], lin1_w [ffn,d], lin2_w [d,ffn]; lin* NULL for the
 * cross-attention-only decoder of SHA_FFN_CRSATTN).  workspace: device floats, gom_match_workspace_floats(...).
 * traj [hi-lo, num_tracks] device. */
typedef struct gom_matcher_layer {
    const float* in_w;
    const float* in_b;
    const float* out_w;
    const float* out_b;
    const float* lin1_w;
    const float* lin1_b;
    const float* lin2_w;
    const float* lin2_b;
} gom_matcher_layer;
long gom_match_workspace_floats(int N, int n_k, int d, int ffn);
/* [device] one launch for a match with hoisted projections: src [N, dim] <- pool[rows], qkv [N, 3 dim] <- proj[rows][0 : 3 dim],
 * qdec [n_k, dim] <- proj[rows[lo .. lo + n_k)][3 dim : 4 dim]. */
int gom_gather_match_f32(const float* pool, int ld_pool, const float* proj, int ld_proj, const int* rows, int N, int lo, int n_k,
                         int dim, float* src, float* qkv, float* qdec, void* stream);
/* gom_match_scores_f32 with the two per-row projections of the raw embeddings hoisted out of the chain: proj [pool rows,
 * ld_proj >= 4 d] = (encoder layer 0 in-projection | decoder layer 0 query projection) of every pool row, computed once per
 * detection by gom_gemm_small_f32 (same bits as in the chain).  proj == NULL: gom_match_scores_f32. */
int gom_match_scores_proj_f32(const float* pool, int ld_pool, const float* proj, int ld_proj, const int* rows,
                              const int* frame_offsets, const int* meta, const float* boxes, const float* decay, int N, int T,
                              int lo, int hi, int num_tracks, const gom_matcher_layer* enc, int n_enc,
                              const gom_matcher_layer* dec, int n_dec, int d, int heads, int ffn, float img_w, float img_h,
                              int with_iou, float max_center_dist, float* workspace, long workspace_floats, float* traj,
                              void* stream);
int gom_match_scores_f32(const float* pool, int ld_pool, const int* rows, const int* frame_offsets, const int* meta,
                         const float* boxes, const float* decay, int N, int T, int lo, int hi, int num_tracks,
                         const gom_matcher_layer* enc, int n_enc, const gom_matcher_layer* dec, int n_dec, int d,
                         int heads, int ffn, float img_w, float img_h, int with_iou, float max_center_dist,
                         float* workspace, long workspace_floats, float* traj, void* stream);
/* [host + device] The per-frame id recurrence of GoMatching.track_frames for all frames of a call behind ONE crossing
 * (tracker_rt.hip; gom_lstmatcher.py:366-564): short-term assignment from precomputed score matrices, long-term match
 * (selection, descriptors, gom_match_scores_f32, LSA, thresholds, id allocation).  Host arrays in, ids out; see the
 * comment above gom_tracker_run in tracker_rt.hip for the layout.  The handle owns its device / pinned scratch. */
void* gom_tracker_create(int test_len, float overlap_thresh, int not_mult_thresh, int use_decay, int with_iou,
                         float max_center_dist, const gom_matcher_layer* enc, int n_enc, const gom_matcher_layer* dec,
                         int n_dec, int d, int heads, int ffn);
void gom_tracker_destroy(void* tracker);
/* [host] hoisted projections (gom_match_scores_proj_f32) for the NEXT gom_tracker_run call only; NULL = none. */
int gom_tracker_set_projections(void* tracker, const float* proj_dev, int ld_proj);

int gom_tracker_run(void* tracker, int F, const int* n, const float* boxes, const int* rows, long* ids, int first_new,
                    long first_real, const float* S, const long* s_off, const float* pool_dev, int ld_pool, float img_w,
                    float img_h, const float* decay_table, long* id_count_io, double* secs, void* stream);
/* The same with one image size PER window frame (frame_wh [F, 2] = width, height): a long-term match then normalises every box
 * of its window by the size of the window's FIRST frame, as the reference does (gom_lstmatcher.py:471 builds every window
 * Instances with full_instances[0].image_size; lstmatcher.py:478-494 divides by it) -- clips that mix resolutions (BOVText). */
int gom_tracker_run_wh(void* tracker, int F, const int* n, const float* boxes, const int* rows, long* ids, int first_new,
                       long first_real, const float* S, const long* s_off, const float* pool_dev, int ld_pool,
                       const float* frame_wh, const float* decay_table, long* id_count_io, double* secs, void* stream);
/* ---- CU-partitioned streams (csrc/stream.hip): the tracker's lane of a multi-GPU step ------------------------------------
 * [host] A HIP stream whose kernels run only on the compute units of cu_mask (`words` x 32 bits, bit i = CU i of the driver's
 * enumeration).  GoMatching.reserve_tracker_cus() gives the tracker stream a few CUs of every XCD and the detector stream the
 * rest, so that the tracker's small dependent launches never queue behind the detector's resident workgroups. */
int gom_stream_create_cu_mask(const unsigned* cu_mask, int words, void** stream_out);
int gom_stream_destroy(void* stream);

/* [host] rectangular assignment, SciPy-compatible tie-breaking (gom_lstmatcher.py:447,549).  Returns the number
 * of assigned pairs (min(nr,nc)) or a negative error. */
int gom_linear_sum_assignment(const double* cost, long nr, long nc, long* row_ind, long* col_ind);

/* Multi-GPU exchange: one launch packs a step's detections into the all-gather buffer [frames, nq + 1, D]
 * fp32, D = feature_dim + 4 + 1 + 2 P + 4 P + P: row 0 of a frame = (count, image height, image width, 0...), rows 1..count =
 * reid | box | score | ctrl | bd | recs, the rest zero.  The re-id rows of frame f start at pool row
 * row_base + sum(counts[0..f)); boxes / scores / ctrl / bd / recs are the nq-padded arrays of gom_detect_post. */
int gom_pack_records_f32(const float* pool, int ld_pool, int row_base, const int* counts, const float* boxes,
                         const float* scores, const float* ctrl, const float* bd, const long* recs, int frames, int nq,
                         int feature_dim, int num_points, float img_h, float img_w, float* out, void* stream);

/* Training of the association head (only `roi_heads` trains, freeze_layers.py:20-37) -- the pieces that are
 * not contractions (those run on gom_gemm_f32 with transposed operands, gomatching_amd/training.py):
 *   gom_relu_backward_f32          dx = y > 0 ? dy : 0
 *   gom_softmax_rows_backward_f32  dS = scale * P * (dP - rowsum(dP * P))                     (attention backward)
 *   gom_asso_ce_f32                detr_asso_loss's per-frame cross entropy with a zero background logit (lstmatcher.py:
 *                                  436-475): per (row, frame) loss and / or dlogits = *grad_scale * (softmax - onehot);
 *                                  gt[row*T + t] = target index inside frame t, n_t = background, < 0 = pair not counted
 *   gom_sigmoid_focal_f32          loss_res's sigmoid focal loss per element (lstmatcher.py:237-268) and / or d loss / d x */
int gom_relu_backward_f32(const float* dy, const float* y, float* dx, long n, void* stream);
int gom_softmax_rows_backward_f32(const float* P, const float* dP, float* dS, long rows, int cols, long ld, float scale,
                                  void* stream);
int gom_asso_ce_f32(const float* logits, int ld, const int* frame_offsets, int num_frames, const int* gt, long rows,
                    float* loss, const float* grad_scale, float* dlogits, void* stream);
int gom_sigmoid_focal_f32(const float* x, const float* target, float alpha, float gamma, long n, float* loss, float* dx,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GOMATCHING_HIP_H_ */
