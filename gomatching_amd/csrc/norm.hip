// LayerNorm (with fused residual add) and channels-last GroupNorm for gfx950.  HBM-bound row ops:
// one wavefront per 256-wide row, 16-byte loads, reductions by cross-lane shuffles only.
#include "common.h"

namespace {

// out[r,:] = LN(x[r,:] + res[r,:]) * gamma + beta ; D = 64*4*VEC floats per row, one wave per row.
// (deformable_transformer.py:272-273, 250-251, 393-394, 403-404, 421-422, 368-369; eps 1e-5)
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ out,
                                                        long rows, float eps) {
    constexpr int D = 256 * VEC;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    f32x4 v[VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const size_t o = (size_t)row * D + (i * 64 + lane) * 4;
        v[i] = *reinterpret_cast<const f32x4*>(x + o);
        if (res) v[i] += *reinterpret_cast<const f32x4*>(res + o);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const f32x4 d = v[i] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int c = (i * 64 + lane) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        *reinterpret_cast<f32x4*>(out + (size_t)row * D + c) = (v[i] - mean) * rstd * g + b;
    }
}

// GroupNorm(32, 256) on [B, HW, 256] channels-last (detection_transformer_wobackbone.py:77-88).
// pass 1: per (b, group) sum and sum of squares in fp64 (atomics on 2*B*32 doubles);
// pass 2: normalise and write straight into the flattened multi-level token buffer.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                       int HW, int rows_per_block) {
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane owns channels 4*lane..4*lane+3 -> group (lane >> 1); a wave strides over rows
    double s = 0.0, q = 0.0;
    for (int r = r0 + wave; r < r1; r += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)b * HW + r) * 256 + lane * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += (double)v[k]; q += (double)v[k] * (double)v[k]; }
    }
    s += __shfl_xor(s, 1, 64);
    q += __shfl_xor(q, 1, 64);
    __shared__ double sh[4][32][2];
    if ((lane & 1) == 0) { sh[wave][lane >> 1][0] = s; sh[wave][lane >> 1][1] = q; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int g = threadIdx.x >> 1, w = threadIdx.x & 1;
        const double t = sh[0][g][w] + sh[1][g][w] + sh[2][g][w] + sh[3][g][w];
        atomicAdd(&stats[((size_t)b * 32 + g) * 2 + w], t);
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out,
                                                       int HW, long out_batch_stride, float eps) {
    const int b = blockIdx.y;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;             // float4 index within the image
    if (idx >= (long)HW * 64) return;
    const int c = (int)(idx & 63) * 4;
    const long r = idx >> 6;
    const int g = c >> 3;
    const double n = (double)HW * 8.0;
    const double mean = stats[((size_t)b * 32 + g) * 2] / n;
    double var = stats[((size_t)b * 32 + g) * 2 + 1] / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float mu = (float)mean;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)b * HW + r) * 256 + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
    *reinterpret_cast<f32x4*>(out + (size_t)b * out_batch_stride + r * 256 + c) = (v - mu) * rstd * ga + be;
}

}  // namespace

extern "C" int gom_layernorm_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                 float* out, long rows, int dim, float eps, void* stream) {
    GOM_CHECK_ARG(x && gamma && beta && out && rows >= 0);
    GOM_CHECK_ARG(dim == 256 || dim == 1024);
    if (rows == 0) return GOM_OK;
    const dim3 grid((unsigned)cdiv(rows, 4));
    if (dim == 256)
        hipLaunchKernelGGL((layernorm_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, residual, gamma, beta,
                           out, rows, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, x, residual, gamma, beta,
                           out, rows, eps);
    return gom_launch_status();
}

extern "C" int gom_groupnorm32_nhwc_f32(const float* x, const float* gamma, const float* beta, double* stats_ws,
                                        float* out, long out_batch_stride, int B, int HW, int channels, float eps,
                                        void* stream) {
    GOM_CHECK_ARG(x && gamma && beta && stats_ws && out);
    GOM_CHECK_ARG(channels == 256 && B > 0 && HW > 0 && out_batch_stride >= (long)HW * 256);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats_ws, 0, sizeof(double) * B * 32 * 2, s);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    const int rows_per_block = 256;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)cdiv(HW, rows_per_block), (unsigned)B), dim3(256), 0, s, x,
                       stats_ws, HW, rows_per_block);
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)cdiv((long)HW * 64, 256), (unsigned)B), dim3(256), 0, s, x,
                       stats_ws, gamma, beta, out, HW, out_batch_stride, eps);
    return gom_launch_status();
}
