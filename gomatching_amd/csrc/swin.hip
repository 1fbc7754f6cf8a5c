// Swin-T backbone glue (third_party/adet/modeling/swin/swin_transformer.py).  The linear layers run on the
// GEMM kernels; what is specific to Swin is data movement around 7x7 windows -- HBM-bound gathers with coalesced
// channel-contiguous 16-byte accesses -- a LayerNorm over any channel count, exact GELU, and the window attention core
// (49 x 49 scores per head with the relative-position bias and the shifted-window mask added before the softmax).
#include "common.h"

namespace {

constexpr int WS = 7, WT = WS * WS;

// ---- LayerNorm over the last dimension, any D <= 2048 with D % 4 == 0 (96 ... 1536): the row in registers, centred
// variance (two reductions) as nn.LayerNorm.  One wave per row; for D <= 128 (Swin stage 0: 96 channels over 900k
// tokens) one HALF-wave per row, so that 24 of 32 lanes work instead of 24 of 64.
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LANES>                                          // lanes per row: 64 or 32
__global__ __launch_bounds__(256) void layernorm_any_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            long rows, int D, float eps) {
    constexpr int RPB = 256 / LANES, NV = LANES == 64 ? 8 : 1;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / LANES;
    const bool live = row < rows;
    const int lane = threadIdx.x % LANES;
    const float* xr = x + (live ? row : 0) * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + LANES * i) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = group_sum<LANES>(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + LANES * i) * 4;
        if (c < D) {
            const f32x4 d = v[i] - mean;
            q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
    }
    const float rstd = rsqrtf(group_sum<LANES>(q) / (float)D + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + LANES * i) * 4;
        if (c < D) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
            *reinterpret_cast<f32x4*>(out + row * D + c) = (v[i] - mean) * rstd * g + b;
        }
    }
}

__global__ __launch_bounds__(256) void gelu_kernel(float* __restrict__ x, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = *reinterpret_cast<f32x4*>(x + i * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = 0.5f * v[k] * (1.f + erff(v[k] * 0.70710678118654752440f));   // nn.GELU (exact)
    *reinterpret_cast<f32x4*>(x + i * 4) = v;
}

// ---- image [B,H,W,4] (normalised, 4th channel 0) -> patch rows [B*Hp*Wp, 64] in (kh, kw, c) order; zero beyond H/W
// (PatchEmbed pads right/bottom to multiples of 4, :475-478).
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H,
                                                       int W, int Hp, int Wp, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // one float4 = one source pixel
    if (i >= total) return;
    const int kw = (int)(i & 3), kh = (int)((i >> 2) & 3);
    long t = i >> 4;
    const int px = (int)(t % Wp);
    t /= Wp;
    const int py = (int)(t % Hp);
    const long b = t / Hp;
    const int y = py * 4 + kh, x = px * 4 + kw;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y < H && x < W) v = *reinterpret_cast<const f32x4*>(img + ((b * H + y) * W + x) * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// ---- tokens [B,H,W,C] -> window rows [B*nWy*nWx*49, C]: pad to multiples of 7 (zeros), cyclic shift by -shift, window
// partition (SwinTransformerBlock.forward :246-262).
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ x, float* __restrict__ out, int H,
                                                            int W, int C4, int Hp, int Wp, int shift, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // float4 index over the window rows
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
    int y = wy * WS + t / WS + shift, xx = wx * WS + t % WS + shift;      // shifted_x[y'] = x[(y' + shift) mod Hp]
    if (y >= Hp) y -= Hp;
    if (xx >= Wp) xx -= Wp;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y < H && xx < W) v = *reinterpret_cast<const f32x4*>(x + (((b * H + y) * W + xx) * C4 + c) * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// ---- out[b,y,x] = shortcut[b,y,x] + window rows at the position the reverse shift / window_reverse / crop put there
// (:264-279).
__global__ __launch_bounds__(256) void window_scatter_add_kernel(const float* __restrict__ win,
                                                                 const float* __restrict__ shortcut,
                                                                 float* __restrict__ out, int H, int W, int C4, int Hp,
                                                                 int Wp, int shift, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // float4 index over [B,H,W,C]
    if (i >= total) return;
    const int c = (int)(i % C4);
    long r = i / C4;
    const int xx = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const long b = r / H;
    int ys = y - shift, xs = xx - shift;                       // x[y] = shifted_x[(y - shift) mod Hp]
    if (ys < 0) ys += Hp;
    if (xs < 0) xs += Wp;
    const int nwx = Wp / WS, nwy = Hp / WS;
    const long row = ((b * nwy + ys / WS) * nwx + xs / WS) * WT + (ys % WS) * WS + xs % WS;
    const f32x4 a = *reinterpret_cast<const f32x4*>(win + (row * C4 + c) * 4);
    const f32x4 s = *reinterpret_cast<const f32x4*>(shortcut + i * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = s + a;
}

// ---- PatchMerging gather (:320-327): [B,H,W,C] -> [B,H2,W2,4C] = cat(x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2])
__global__ __launch_bounds__(256) void patch_merge_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                          int C4, int H2, int W2, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // float4 index over [B,H2,W2,4,C4]
    if (i >= total) return;
    const int c = (int)(i % C4);
    long r = i / C4;
    const int part = (int)(r & 3);
    r >>= 2;
    const int x2 = (int)(r % W2);
    r /= W2;
    const int y2 = (int)(r % H2);
    const long b = r / H2;
    const int y = 2 * y2 + (part & 1), xx = 2 * x2 + (part >> 1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y < H && xx < W) v = *reinterpret_cast<const f32x4*>(x + (((b * H + y) * W + xx) * C4 + c) * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

// ---- window attention core (WindowAttention.forward :139-165): one wave per (window, head); lane = query token (49 of
// 64), q row and output row in registers, the window's K and V for this head in LDS (broadcast reads), scores
// q.k * scale + relative-position bias [+ shift mask of the window's position], exact two-pass softmax.
__global__ __launch_bounds__(256) void window_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               const float* __restrict__ bias,      // [heads][49][49]
                                                               const float* __restrict__ mask,      // [nW][49][49] or null
                                                               int heads, int C, int nW, long total, float scale) {
    constexpr int HD = 32;
    __shared__ __attribute__((aligned(16))) float smem[4][2][WT * HD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + wave;            // (window, head), head fastest
    const bool on = item < total;
    const long win = on ? item / heads : 0;
    const int h = on ? (int)(item % heads) : 0;
    const float* base = qkv + win * WT * 3 * C + h * HD;
    float* Ks = smem[wave][0];
    float* Vs = smem[wave][1];
    for (int u = lane; u < WT * (HD / 4); u += 64) {
        const int r = u / (HD / 4), d4 = (u % (HD / 4)) * 4;
        *reinterpret_cast<f32x4*>(Ks + r * HD + d4) = *reinterpret_cast<const f32x4*>(base + (long)r * 3 * C + C + d4);
        *reinterpret_cast<f32x4*>(Vs + r * HD + d4) = *reinterpret_cast<const f32x4*>(base + (long)r * 3 * C + 2 * C + d4);
    }
    const int i = lane < WT ? lane : WT - 1;                   // idle lanes recompute the last row (no divergence)
    f32x4 qv[HD / 4];
#pragma unroll
    for (int d = 0; d < HD / 4; ++d) qv[d] = *reinterpret_cast<const f32x4*>(base + (long)i * 3 * C + 4 * d) * scale;
    __builtin_amdgcn_s_waitcnt(0xC07F);                        // the K / V slab is wave-private
    __builtin_amdgcn_wave_barrier();
    const float* brow = bias + ((long)h * WT + i) * WT;
    const float* mrow = mask ? mask + ((win % nW) * WT + i) * WT : nullptr;
    auto score = [&](int j) {
        const float* kr = Ks + j * HD;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int d = 0; d < HD / 4; d += 2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(kr + 4 * d);
            const f32x4 b = *reinterpret_cast<const f32x4*>(kr + 4 * d + 4);
            s0 = fmaf(qv[d][0], a[0], s0); s0 = fmaf(qv[d][1], a[1], s0);
            s0 = fmaf(qv[d][2], a[2], s0); s0 = fmaf(qv[d][3], a[3], s0);
            s1 = fmaf(qv[d + 1][0], b[0], s1); s1 = fmaf(qv[d + 1][1], b[1], s1);
            s1 = fmaf(qv[d + 1][2], b[2], s1); s1 = fmaf(qv[d + 1][3], b[3], s1);
        }
        float s = s0 + s1 + brow[j];
        if (mrow) s += mrow[j];
        return s;
    };
    float mx = -INFINITY;
    for (int j = 0; j < WT; ++j) mx = fmaxf(mx, score(j));
    f32x4 acc[HD / 4];
#pragma unroll
    for (int d = 0; d < HD / 4; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sum = 0.f;
    for (int j = 0; j < WT; ++j) {
        const float e = expf(score(j) - mx);
        sum += e;
        const float* vr = Vs + j * HD;
#pragma unroll
        for (int d = 0; d < HD / 4; ++d) acc[d] += *reinterpret_cast<const f32x4*>(vr + 4 * d) * e;
    }
    if (on && lane < WT) {
        const float inv = 1.f / sum;
        float* o = out + (win * WT + lane) * C + h * HD;
#pragma unroll
        for (int d = 0; d < HD / 4; ++d) *reinterpret_cast<f32x4*>(o + 4 * d) = acc[d] * inv;
    }
}

}  // namespace

#define SWIN_GRID(n) dim3((unsigned)cdiv((n), 256)), dim3(256), 0, (hipStream_t)stream

extern "C" int gom_layernorm_any_f32(const float* x, const float* gamma, const float* beta, float* out, long rows, int dim,
                                     float eps, void* stream) {
    GOM_CHECK_ARG(x && gamma && beta && out && rows >= 0 && dim > 0 && dim <= 2048 && (dim % 4) == 0);
    if (rows == 0) return GOM_OK;
    if (dim <= 128)
        hipLaunchKernelGGL(layernorm_any_kernel<32>, dim3((unsigned)cdiv(rows, 8)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                           beta, out, rows, dim, eps);
    else
        hipLaunchKernelGGL(layernorm_any_kernel<64>, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                           beta, out, rows, dim, eps);
    return gom_launch_status();
}

extern "C" int gom_gelu_f32(float* x, long n, void* stream) {
    GOM_CHECK_ARG(x && n >= 0 && (n % 4) == 0);
    if (n == 0) return GOM_OK;
    hipLaunchKernelGGL(gelu_kernel, SWIN_GRID(n / 4), x, n / 4);
    return gom_launch_status();
}

extern "C" int gom_swin_patchify_f32(const float* img, float* out, int B, int H, int W, void* stream) {
    GOM_CHECK_ARG(img && out && B > 0 && H > 0 && W > 0);
    const int Hp = (H + 3) / 4, Wp = (W + 3) / 4;
    const long total = (long)B * Hp * Wp * 16;
    hipLaunchKernelGGL(patchify_kernel, SWIN_GRID(total), img, out, B, H, W, Hp, Wp, total);
    return gom_launch_status();
}

extern "C" int gom_swin_window_gather_f32(const float* x, float* out, int B, int H, int W, int C, int shift, void* stream) {
    GOM_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && shift >= 0 && shift < WS);
    const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
    const long total = (long)B * Hp * Wp * (C / 4);
    hipLaunchKernelGGL(window_gather_kernel, SWIN_GRID(total), x, out, H, W, C / 4, Hp, Wp, shift, total);
    return gom_launch_status();
}

extern "C" int gom_swin_window_scatter_add_f32(const float* windows, const float* shortcut, float* out, int B, int H, int W,
                                               int C, int shift, void* stream) {
    GOM_CHECK_ARG(windows && shortcut && out && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && shift >= 0 &&
                  shift < WS);
    const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
    const long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(window_scatter_add_kernel, SWIN_GRID(total), windows, shortcut, out, H, W, C / 4, Hp, Wp, shift, total);
    return gom_launch_status();
}

extern "C" int gom_swin_patch_merge_f32(const float* x, float* out, int B, int H, int W, int C, void* stream) {
    GOM_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0);
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const long total = (long)B * H2 * W2 * 4 * (C / 4);
    hipLaunchKernelGGL(patch_merge_kernel, SWIN_GRID(total), x, out, H, W, C / 4, H2, W2, total);
    return gom_launch_status();
}

extern "C" int gom_swin_window_attention_f32(const float* qkv, float* out, const float* bias, const float* mask,
                                             long num_windows, int windows_per_image, int heads, int C, void* stream) {
    GOM_CHECK_ARG(qkv && out && bias && num_windows >= 0 && windows_per_image > 0 && heads > 0 && C == heads * 32);
    GOM_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0);
    if (num_windows == 0) return GOM_OK;
    const long total = num_windows * heads;
    hipLaunchKernelGGL(window_attention_kernel, dim3((unsigned)cdiv(total, 4)), dim3(256), 0, (hipStream_t)stream, qkv, out,
                       bias, mask, heads, C, windows_per_image, total, 1.0f / sqrtf(32.f));
    return gom_launch_status();
}
