// ViTAEv2-S backbone glue (third_party/adet/modeling/vitae_v2/{vitae_v2,ReductionCell,NormalCell,window,
// token_transformer}.py).  Linear layers, the dense 3x3 convolutions and the attention products run on the GEMM kernels;
// what is specific to ViTAE lives here: the dilated strided convolutions of the pyramid reduction module as an im2col
// feeding those GEMMs, the grouped 3x3 convolutions of the parallel convolution branch (4 or 16 channels per group: an
// HBM-bound direct kernel, BatchNorm folded, SiLU fused), centred 7x7 window gather / crop, the window attention core
// for 64- and 128-wide heads (no position bias in this model), and the row softmax of the full attention.
#include "common.h"

namespace {

constexpr int WS = 7, WT = WS * WS;

__device__ __forceinline__ float silu(float v) { return v / (1.f + expf(-v)); }

// ---- x [B,H,W,C] -> rows [B*OH*OW, ldo] in (kh, kw, c) order, zero outside the map and in the columns >= KH*KW*C
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int dil, int OH, int OW, int ldo4,
                                                     long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // float4 index over [M, ldo/4]
    if (i >= total) return;
    const int col = (int)(i % ldo4) * 4;
    long r = i / ldo4;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const long b = r / OH;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int tap = col / C, c = col % C;
    if (tap < KH * KW) {
        const int kh = tap / KW, kw = tap % KW;
        const int iy = oy * stride - pad + kh * dil, ix = ox * stride - pad + kw * dil;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(x + ((b * H + iy) * W + ix) * C + c);
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// ---- grouped 3x3 convolution, padding 1, NHWC, weights [3][3][Cout][CG] (CG = Cin / groups input channels per group);
// y = act(conv * scale + shift) [+ R].  One thread per (pixel, output channel): neighbouring threads share the group's
// input pixels (broadcast in the vector cache) and write coalesced.
template <int CG>
__global__ __launch_bounds__(256) void grouped_conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ R, float* __restrict__ y, int H, int W,
                                                              int Cin, int Cout, int cout_g, int stride, int OH, int OW, int act,
                                                              long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % Cout);
    long r = i / Cout;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const long b = r / OH;
    const int g = co / cout_g;
    const float* wr = w + (long)co * CG;                      // [3][3][Cout][CG]: a wave's lanes read contiguous weights
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy * stride - 1 + kh;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = ox * stride - 1 + kw;
            if (ix < 0 || ix >= W) continue;
            const float* xp = x + ((b * H + iy) * W + ix) * Cin + g * CG;
            const float* wp = wr + (long)(kh * 3 + kw) * Cout * CG;
#pragma unroll
            for (int c = 0; c < CG; c += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xp + c), q = *reinterpret_cast<const f32x4*>(wp + c);
                acc = fmaf(a[0], q[0], acc); acc = fmaf(a[1], q[1], acc);
                acc = fmaf(a[2], q[2], acc); acc = fmaf(a[3], q[3], acc);
            }
        }
    }
    float v = acc * (scale ? scale[co] : 1.f) + shift[co];
    if (act == 3) v = silu(v);
    if (R) v += R[i];
    y[i] = v;
}

__global__ __launch_bounds__(256) void silu_kernel(float* __restrict__ x, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = *reinterpret_cast<f32x4*>(x + i * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = silu(v[k]);
    *reinterpret_cast<f32x4*>(x + i * 4) = v;
}

// ---- tokens [B,H,W,C] -> window rows [B*nWy*nWx*49, C] of the grid zero-padded to multiples of 7 with the padding split
// top/bottom and left/right (ReductionCell.py:147-156, NormalCell.py:160-165)
__global__ __launch_bounds__(256) void window_gather_centred_kernel(const float* __restrict__ x, float* __restrict__ out, int H,
                                                                    int W, int C4, int Hp, int Wp, int top, int left,
                                                                    long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    long r = i / C4;
    const int t = (int)(r % WT);
    r /= WT;
    const int nwx = Wp / WS, nwy = Hp / WS;
    const int wx = (int)(r % nwx);
    r /= nwx;
    const int wy = (int)(r % nwy);
    const long b = r / nwy;
    const int y = wy * WS + t / WS - top, xx = wx * WS + t % WS - left;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f32x4*>(x + (((b * H + y) * W + xx) * C4 + c) * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// window rows -> tokens [B,H,W,C] (window reverse + crop), plus up to two addends of the token shape
__global__ __launch_bounds__(256) void window_crop_kernel(const float* __restrict__ win, const float* __restrict__ R1,
                                                          const float* __restrict__ R2, float* __restrict__ out, int H, int W,
                                                          int C4, int Hp, int Wp, int top, int left, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // float4 index over [B,H,W,C4]
    if (i >= total) return;
    const int c = (int)(i % C4);
    long r = i / C4;
    const int xx = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const long b = r / H;
    const int py = y + top, px = xx + left;
    const int nwx = Wp / WS, nwy = Hp / WS;
    const long row = ((b * nwy + py / WS) * nwx + px / WS) * WT + (py % WS) * WS + px % WS;
    f32x4 v = *reinterpret_cast<const f32x4*>(win + (row * C4 + c) * 4);
    if (R1) v += *reinterpret_cast<const f32x4*>(R1 + i * 4);
    if (R2) v += *reinterpret_cast<const f32x4*>(R2 + i * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// ---- window attention core for HD-wide heads (window.py:92-124 without mask): one wave per (window, head); lane = query
// token; the 49 scores live in registers, K and V of the head in LDS (broadcast reads), q streamed in 32-wide chunks.
template <int HD>
__global__ __launch_bounds__(64) void window_attention_wide_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                   int heads, int C, long total, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[WT * HD];
    __shared__ __attribute__((aligned(16))) float Vs[WT * HD];
    const int lane = threadIdx.x;
    const long item = blockIdx.x;                              // (window, head), head fastest
    const long win = item / heads;
    const int h = (int)(item % heads);
    const float* base = qkv + win * WT * 3 * C + h * HD;
    for (int u = lane; u < WT * (HD / 4); u += 64) {
        const int r = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
        *reinterpret_cast<f32x4*>(Ks + r * HD + d4) = *reinterpret_cast<const f32x4*>(base + (long)r * 3 * C + C + d4);
        *reinterpret_cast<f32x4*>(Vs + r * HD + d4) = *reinterpret_cast<const f32x4*>(base + (long)r * 3 * C + 2 * C + d4);
    }
    __syncthreads();
    const int i = lane < WT ? lane : WT - 1;
    float s[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) s[j] = 0.f;
    for (int c0 = 0; c0 < HD; c0 += 32) {
        f32x4 qv[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) qv[d] = *reinterpret_cast<const f32x4*>(base + (long)i * 3 * C + c0 + 4 * d) * scale;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const float* kr = Ks + j * HD + c0;
            float a = s[j];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 4 * d);
                a = fmaf(qv[d][0], k4[0], a); a = fmaf(qv[d][1], k4[1], a);
                a = fmaf(qv[d][2], k4[2], a); a = fmaf(qv[d][3], k4[3], a);
            }
            s[j] = a;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < WT; ++j) mx = fmaxf(mx, s[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        s[j] = expf(s[j] - mx);
        sum += s[j];
    }
    const float inv = 1.f / sum;
    float* o = out + (win * WT + i) * C + h * HD;
    for (int c0 = 0; c0 < HD; c0 += 32) {
        f32x4 acc[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const float* vr = Vs + j * HD + c0;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] += *reinterpret_cast<const f32x4*>(vr + 4 * d) * s[j];
        }
        if (lane < WT) {
#pragma unroll
            for (int d = 0; d < 8; ++d) *reinterpret_cast<f32x4*>(o + c0 + 4 * d) = acc[d] * inv;
        }
    }
}

// ---- in-place softmax(row * scale) over rows of up to 256*32 columns: one workgroup per row, the row in registers
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int cols, long ld, float scale) {
    __shared__ float red[4];
    float* row = x + (long)blockIdx.x * ld;
    float v[32];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = threadIdx.x + 256 * k;
        v[k] = c < cols ? row[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v[k] = expf(v[k] - mx);                              // exp(-inf) = 0 beyond the row
        sum += v[k];
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < cols) row[c] = v[k] * inv;
    }
}

// ---- out[c][r] = x[r * ld + c]  (V of one head, [N, HD] strided -> [HD, N]: the P.V product reads it as a weight matrix)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols,
                                                        long ld, long ldo) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < rows && c < cols) ? x[(long)r * ld + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) out[(long)c * ldo + r] = tile[tx][k];
    }
}

}  // namespace

#define V_GRID(n) dim3((unsigned)cdiv((n), 256)), dim3(256), 0, (hipStream_t)stream

extern "C" int gom_im2col_nhwc_f32(const float* x, float* out, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                                   int dilation, int ldo, void* stream) {
    GOM_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 &&
                  dilation > 0 && ldo >= KH * KW * C && (ldo % 4) == 0);
    const int OH = (H + 2 * pad - dilation * (KH - 1) - 1) / stride + 1, OW = (W + 2 * pad - dilation * (KW - 1) - 1) / stride + 1;
    GOM_CHECK_ARG(OH > 0 && OW > 0);
    const long total = (long)B * OH * OW * (ldo / 4);
    hipLaunchKernelGGL(im2col_kernel, V_GRID(total), x, out, H, W, C, KH, KW, stride, pad, dilation, OH, OW, ldo / 4, total);
    return gom_launch_status();
}

extern "C" int gom_grouped_conv3x3_nhwc_f32(const float* x, const float* w, const float* scale, const float* shift,
                                            const float* R, int act, float* y, int B, int H, int W, int Cin, int Cout,
                                            int groups, int stride, void* stream) {
    GOM_CHECK_ARG(x && w && shift && y && B > 0 && H > 0 && W > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0);
    GOM_CHECK_ARG((stride == 1 || stride == 2) && (act == 0 || act == 3));
    const int cg = Cin / groups, OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    GOM_CHECK_ARG(cg == 4 || cg == 16);
    const long total = (long)B * OH * OW * Cout;
    if (cg == 4)
        hipLaunchKernelGGL(grouped_conv3x3_kernel<4>, V_GRID(total), x, w, scale, shift, R, y, H, W, Cin, Cout, Cout / groups,
                           stride, OH, OW, act, total);
    else
        hipLaunchKernelGGL(grouped_conv3x3_kernel<16>, V_GRID(total), x, w, scale, shift, R, y, H, W, Cin, Cout, Cout / groups,
                           stride, OH, OW, act, total);
    return gom_launch_status();
}

extern "C" int gom_silu_f32(float* x, long n, void* stream) {
    GOM_CHECK_ARG(x && n >= 0 && (n % 4) == 0);
    if (n == 0) return GOM_OK;
    hipLaunchKernelGGL(silu_kernel, V_GRID(n / 4), x, n / 4);
    return gom_launch_status();
}

extern "C" int gom_vitae_window_gather_f32(const float* x, float* out, int B, int H, int W, int C, void* stream) {
    GOM_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0);
    const int td = (WS - H % WS) % WS, lr = (WS - W % WS) % WS;
    const long total = (long)B * (H + td) * (W + lr) * (C / 4);
    hipLaunchKernelGGL(window_gather_centred_kernel, V_GRID(total), x, out, H, W, C / 4, H + td, W + lr, td / 2, lr / 2, total);
    return gom_launch_status();
}

extern "C" int gom_vitae_window_crop_f32(const float* windows, const float* R1, const float* R2, float* out, int B, int H, int W,
                                         int C, void* stream) {
    GOM_CHECK_ARG(windows && out && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0);
    const int td = (WS - H % WS) % WS, lr = (WS - W % WS) % WS;
    const long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(window_crop_kernel, V_GRID(total), windows, R1, R2, out, H, W, C / 4, H + td, W + lr, td / 2, lr / 2, total);
    return gom_launch_status();
}

extern "C" int gom_vitae_window_attention_f32(const float* qkv, float* out, long num_windows, int heads, int C, void* stream) {
    GOM_CHECK_ARG(qkv && out && num_windows >= 0 && heads > 0 && C % heads == 0 && (C / heads == 64 || C / heads == 128));
    GOM_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0);
    if (num_windows == 0) return GOM_OK;
    const long total = num_windows * heads;
    GOM_CHECK_ARG(total < (1L << 31));
    const int hd = C / heads;
    if (hd == 64)
        hipLaunchKernelGGL(window_attention_wide_kernel<64>, dim3((unsigned)total), dim3(64), 0, (hipStream_t)stream, qkv, out,
                           heads, C, total, 1.0f / sqrtf(64.f));
    else
        hipLaunchKernelGGL(window_attention_wide_kernel<128>, dim3((unsigned)total), dim3(64), 0, (hipStream_t)stream, qkv, out,
                           heads, C, total, 1.0f / sqrtf(128.f));
    return gom_launch_status();
}

extern "C" int gom_softmax_rows_scaled_f32(float* x, long rows, int cols, long ld, float scale, void* stream) {
    GOM_CHECK_ARG(x && rows >= 0 && cols > 0 && cols <= 8192 && ld >= cols && rows < (1L << 31));
    if (rows == 0) return GOM_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, cols, ld, scale);
    return gom_launch_status();
}

extern "C" int gom_transpose_f32(const float* x, float* out, int rows, int cols, long ld, long ldo, void* stream) {
    GOM_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols && ldo >= rows);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), dim3(256), 0,
                       (hipStream_t)stream, x, out, rows, cols, ld, ldo);
    return gom_launch_status();
}
