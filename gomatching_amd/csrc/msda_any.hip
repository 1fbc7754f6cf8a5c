// General form of the native op `adet._C` binds: multi-scale deformable attention for ANY heads / channels / levels /
// points in fp32 or fp64, forward and backward.
//
//   reference: third_party/adet/layers/csrc/vision.cpp:52-55 (the two bound callables), ms_deform_attn_cuda.cu:20-80 (forward
//   host side: shapes from the tensors, AT_DISPATCH_FLOATING_TYPES on the value dtype) and :83-156 (backward),
//   ms_deform_im2col_cuda.cuh:237-299 (forward kernel: one thread per output element), :301-921 (backward: six kernel
//   variants chosen by channel count, block-level shared-memory reductions, atomics on grad_value).
//
// The hot path never comes here: every shipped config is 8 heads x 32 channels x 4 levels x 4 points in fp32, which msda.hip
// serves (lane-distributed, fused with the location / softmax arithmetic).  This file is what makes the boundary complete --
// a caller with another shape, fp64 tensors or a need for gradients gets an answer instead of an error.
//
// Backward on a 64-wide wavefront: ONE WAVE owns one (batch, query, head); its lanes are the channels.  For each (level,
// point) a lane gathers its channel of the four corners, scatters its share of grad_value with a hardware float atomic (the
// only place two waves can meet), and the three per-sample sums -- d/d(attention weight), d/d(x), d/d(y), each a dot product
// over the channels -- are wave reductions; lane 0 stores them.  No shared memory, no block-size-specialised variants, no
// atomics on the location / weight gradients (so those two outputs are deterministic and need no zero fill).
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ T wave_sum_t(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// where a sample falls on its level: corner offsets (in rows of the level), corner validity, bilinear weights
template <typename T>
struct Sample {
    long r00, r01, r10, r11;      // token index of each corner inside the level (valid ones only are used)
    bool ok, k00, k01, k10, k11;
    T fx, fy;
};

template <typename T>
__device__ __forceinline__ Sample<T> locate(T x, T y, int H, int W) {
    Sample<T> s;
    const T xi = x * (T)W - (T)0.5, yi = y * (T)H - (T)0.5;
    s.ok = yi > (T)-1 && xi > (T)-1 && yi < (T)H && xi < (T)W;          // ms_deform_im2col_cuda.cuh:284 (the same window)
    const T x0f = floor(xi), y0f = floor(yi);
    s.fx = xi - x0f;
    s.fy = yi - y0f;
    const long x0 = (long)x0f, y0 = (long)y0f;
    const bool xa = x0 >= 0 && x0 <= W - 1, xb = x0 + 1 >= 0 && x0 + 1 <= W - 1;
    const bool ya = y0 >= 0 && y0 <= H - 1, yb = y0 + 1 >= 0 && y0 + 1 <= H - 1;
    s.k00 = s.ok && ya && xa;
    s.k01 = s.ok && ya && xb;
    s.k10 = s.ok && yb && xa;
    s.k11 = s.ok && yb && xb;
    s.r00 = y0 * W + x0;
    s.r01 = s.r00 + 1;
    s.r10 = s.r00 + W;
    s.r11 = s.r10 + 1;
    return s;
}

// forward: one thread per output element, channels fastest (a head's corner row is read by consecutive lanes)
template <typename T>
__global__ __launch_bounds__(256) void msda_any_fwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ starts, const T* __restrict__ loc,
                                                           const T* __restrict__ attn, T* __restrict__ out, long total, int S,
                                                           int M, int D, int L, int Q, int P) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % D);
    const long bqm = i / D;
    const int m = (int)(bqm % M);
    const long b = bqm / ((long)M * Q);
    const T* vb = value + b * (long)S * M * D + (long)m * D + d;
    const long row = (long)M * D;
    const T* lp = loc + bqm * L * P * 2;
    const T* ap = attn + bqm * L * P;
    T acc = 0;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const T* vl = vb + starts[l] * row;
        for (int p = 0; p < P; ++p) {
            const Sample<T> s = locate<T>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W);
            const T a = ap[l * P + p];
            const T v00 = s.k00 ? vl[s.r00 * row] : (T)0, v01 = s.k01 ? vl[s.r01 * row] : (T)0;
            const T v10 = s.k10 ? vl[s.r10 * row] : (T)0, v11 = s.k11 ? vl[s.r11 * row] : (T)0;
            const T top = v00 + s.fx * (v01 - v00), bot = v10 + s.fx * (v11 - v10);
            acc += a * (top + s.fy * (bot - top));
        }
    }
    out[i] = acc;
}

// backward: one wave per (batch, query, head), lanes = channels (in rounds of 64 when a head is wider)
template <typename T>
__global__ __launch_bounds__(256) void msda_any_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ value,
                                                           const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
                                                           const T* __restrict__ loc, const T* __restrict__ attn,
                                                           T* __restrict__ gvalue, T* __restrict__ gloc, T* __restrict__ gattn,
                                                           long items, int S, int M, int D, int L, int Q, int P) {
    const int lane = threadIdx.x & 63;
    const long bqm = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bqm >= items) return;
    const int m = (int)(bqm % M);
    const long b = bqm / ((long)M * Q);
    const long row = (long)M * D;
    const long base = b * (long)S * row + (long)m * D;
    const T* lp = loc + bqm * L * P * 2;
    const T* ap = attn + bqm * L * P;
    const T* gp = gout + bqm * D;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long lbase = base + starts[l] * row;
        for (int p = 0; p < P; ++p) {
            const Sample<T> s = locate<T>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W);
            const T a = ap[l * P + p];
            T s_w = 0, s_x = 0, s_y = 0;
            if (s.ok) {                                           // wave-uniform
                for (int d = lane; d < D; d += 64) {
                    const T g = gp[d];
                    const long o00 = lbase + s.r00 * row + d, o01 = lbase + s.r01 * row + d;
                    const long o10 = lbase + s.r10 * row + d, o11 = lbase + s.r11 * row + d;
                    const T v00 = s.k00 ? value[o00] : (T)0, v01 = s.k01 ? value[o01] : (T)0;
                    const T v10 = s.k10 ? value[o10] : (T)0, v11 = s.k11 ? value[o11] : (T)0;
                    const T hx = (T)1 - s.fx, hy = (T)1 - s.fy;
                    const T ga = g * a;
                    if (s.k00) unsafeAtomicAdd(gvalue + o00, hy * hx * ga);
                    if (s.k01) unsafeAtomicAdd(gvalue + o01, hy * s.fx * ga);
                    if (s.k10) unsafeAtomicAdd(gvalue + o10, s.fy * hx * ga);
                    if (s.k11) unsafeAtomicAdd(gvalue + o11, s.fy * s.fx * ga);
                    s_w += g * (hy * (hx * v00 + s.fx * v01) + s.fy * (hx * v10 + s.fx * v11));
                    s_x += ga * (hy * (v01 - v00) + s.fy * (v11 - v10));
                    s_y += ga * (hx * (v10 - v00) + s.fx * (v11 - v01));
                }
                s_w = wave_sum_t(s_w);
                s_x = wave_sum_t(s_x);
                s_y = wave_sum_t(s_y);
            }
            if (lane == 0) {
                gattn[bqm * L * P + l * P + p] = s_w;
                gloc[(bqm * L * P + l * P + p) * 2] = s_x * (T)W;          // locations are normalised: d(x_im)/d(x) = W
                gloc[(bqm * L * P + l * P + p) * 2 + 1] = s_y * (T)H;
            }
        }
    }
}

template <typename T>
int forward_t(const void* value, const int64_t* shapes, const int64_t* starts, const void* loc, const void* attn, void* out,
              int B, int S, int M, int D, int L, int Q, int P, hipStream_t st) {
    const long total = (long)B * Q * M * D;
    hipLaunchKernelGGL((msda_any_fwd_kernel<T>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, (const T*)value, shapes,
                       starts, (const T*)loc, (const T*)attn, (T*)out, total, S, M, D, L, Q, P);
    return gom_launch_status();
}

template <typename T>
int backward_t(const void* gout, const void* value, const int64_t* shapes, const int64_t* starts, const void* loc,
               const void* attn, void* gvalue, void* gloc, void* gattn, int B, int S, int M, int D, int L, int Q, int P,
               hipStream_t st) {
    if (hipMemsetAsync(gvalue, 0, (size_t)B * S * M * D * sizeof(T), st) != hipSuccess) return gom_launch_status();
    const long items = (long)B * Q * M;
    hipLaunchKernelGGL((msda_any_bwd_kernel<T>), dim3((unsigned)cdiv(items, 4)), dim3(256), 0, st, (const T*)gout,
                       (const T*)value, shapes, starts, (const T*)loc, (const T*)attn, (T*)gvalue, (T*)gloc, (T*)gattn, items, S,
                       M, D, L, Q, P);
    return gom_launch_status();
}

bool shape_ok(int B, int S, int M, int D, int L, int Q, int P) {
    return B > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Q > 0 && P > 0 && (long)B * Q * M * L * P < (1L << 40);
}

}  // namespace

extern "C" int gom_ms_deform_attn_forward_any(int dtype, const void* value, const int64_t* spatial_shapes,
                                              const int64_t* level_start_index, const void* sampling_loc,
                                              const void* attn_weight, void* output, int batch, int spatial_size,
                                              int num_heads, int channels, int num_levels, int num_query, int num_point,
                                              void* stream) {
    GOM_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && output);
    GOM_CHECK_ARG(shape_ok(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point));
    if (dtype == GOM_DTYPE_F32)
        return forward_t<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output, batch, spatial_size,
                                num_heads, channels, num_levels, num_query, num_point, (hipStream_t)stream);
    if (dtype == GOM_DTYPE_F64)
        return forward_t<double>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output, batch,
                                 spatial_size, num_heads, channels, num_levels, num_query, num_point, (hipStream_t)stream);
    return GOM_ERR_UNSUPPORTED;                                   // the reference's dispatch macro throws on half / bfloat16 too
}

extern "C" int gom_ms_deform_attn_backward(int dtype, const void* value, const int64_t* spatial_shapes,
                                           const int64_t* level_start_index, const void* sampling_loc,
                                           const void* attn_weight, const void* grad_output, void* grad_value,
                                           void* grad_sampling_loc, void* grad_attn_weight, int batch, int spatial_size,
                                           int num_heads, int channels, int num_levels, int num_query, int num_point,
                                           void* stream) {
    GOM_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && grad_output);
    GOM_CHECK_ARG(grad_value && grad_sampling_loc && grad_attn_weight);
    GOM_CHECK_ARG(shape_ok(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point));
    if (dtype == GOM_DTYPE_F32)
        return backward_t<float>(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_value,
                                 grad_sampling_loc, grad_attn_weight, batch, spatial_size, num_heads, channels, num_levels,
                                 num_query, num_point, (hipStream_t)stream);
    if (dtype == GOM_DTYPE_F64)
        return backward_t<double>(grad_output, value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_value,
                                  grad_sampling_loc, grad_attn_weight, batch, spatial_size, num_heads, channels, num_levels,
                                  num_query, num_point, (hipStream_t)stream);
    return GOM_ERR_UNSUPPORTED;
}
