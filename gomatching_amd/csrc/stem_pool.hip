// ResNet stem as ONE launch: conv 7x7 / stride 2 / pad 3 (4 -> 64 channels, NHWC4 input) + folded BatchNorm + ReLU + max-pool
// 3x3 / stride 2 / pad 1 (reference: detectron2 BasicStem, adet/modeling/backbone via build_resnet_backbone).
//
// As two launches the convolution writes its [B, H/2, W/2, 64] output (910 MB for 8 frames of 1000 x 1778) and the pool reads it
// straight back: 1.8 GB of HBM traffic for 228 MB of result.  Here a workgroup owns a 3 x 8 patch of POOLED pixels: it computes
// the 7 x 17 patch of convolution outputs under it (119 of the tile's 128 rows; one halo row and column are recomputed by the
// neighbours, 24 % more MFMA work on a launch that is far from the MFMA roof), applies scale / shift / ReLU, stages the patch in
// LDS and writes only the 24 x 64 maxima.
//
// The arithmetic of every convolution output is the tile kernel's (gemm_f16x3.hip, <128, 64, 7, 7>): the same f16x3 split of the
// activations, the same k-tile order, the same three plane products per k-step in the same order, the same fma epilogue -- so
// the result equals conv2d + max-pool bit for bit (tests/test_stem_pool_gpu.py), and the same fp16 range contract / *flag holds.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;                            // k-tile: eight taps of four channels
constexpr int ROW_BYTES = 80;                     // 32 fp16 + 16 B pad (odd number of 16-byte slots per row)
constexpr int BM = 128, BN = 64;                  // tile: 128 convolution pixels (119 used) x 64 channels
constexpr int KS = 7, TAPS = KS * KS;             // 49 taps x 4 channels = K 196 (weights zero-padded to ldw)
constexpr int PTH = 3, PTW = 8;                   // pooled pixels per workgroup
constexpr int TH = 2 * PTH + 1, TW = 2 * PTW + 1; // convolution patch under them: 7 x 17
constexpr int PITCH = BN + 4;                     // staged patch row (floats)
constexpr int A_PLANE = BM * ROW_BYTES, W_PLANE = BN * ROW_BYTES;
constexpr int LDS_LOOP = 2 * (((2 * (TH - 1) + KS) * (2 * (TW - 1) + KS) + 1) * 8 + 16) + 2 * W_PLANE, LDS_EPI = BM * PITCH * 4;
constexpr int LDS_BYTES = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;

struct StemArgs {
    const float* X;                               // [B, H, W, 4]
    const unsigned short* Wp;                     // [2][64][ldw] fp16 planes (rows pre-scaled, gom_split_f16x2)
    const float* wscale;                          // [64] inverse row scales
    const float* scale;                           // [64] folded BatchNorm scale (may be null)
    const float* shift;                           // [64] folded BatchNorm shift (may be null)
    float* Y;                                     // [B, PH, PW, 64]
    int* flag;
    long w_plane_stride;
    int H, W, OH, OW, PH, PW, ldw, tiles_h, tiles_w;
};

__device__ __forceinline__ void split2(float x, float y, unsigned int& q0, unsigned int& q1) { gom_split2_f16(x, y, q0, q1); }
__device__ __forceinline__ void split4(const f32x4 v, u32x2& p0, u32x2& p1) {
    unsigned int a0, a1, b0, b1;
    split2(v[0], v[1], a0, a1);
    split2(v[2], v[3], b0, b1);
    p0 = u32x2{a0, b0};
    p1 = u32x2{a1, b1};
}

__global__ __launch_bounds__(256, 3) void stem_pool_kernel(const StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                 // 2 x 2 waves: 64 rows x 32 columns each

    int bid = blockIdx.x;
    const int tw = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int th = bid % p.tiles_h, b = bid / p.tiles_h;
    const int ph0 = th * PTH, pw0 = tw * PTW;
    const int oh0 = 2 * ph0 - 1, ow0 = 2 * pw0 - 1;          // convolution pixel of patch position (0, 0)

    constexpr unsigned RANGE = 0x80000000u, INVALID = 0xC0000000u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (int)RANGE, 0x00020000);

    // ---- the INPUT patch under the tile, staged once: 19 x 39 pixels x 4 channels as two fp16 planes (8 bytes per pixel and
    // plane; out-of-image pixels and one spare slot hold zeros).  The A operand of every tap is read straight from it: before,
    // every k-tile gathered its 128 x 8 taps from global memory (16 bytes per lane, ~16 lines per wave-instruction, 28
    // instructions per wave and tile: the texture-address unit, which pays per line, was the busy unit -- 6.4k of its cycles per
    // tile against 2.7k of MFMA issue), split them and stored them to LDS again.  Same values, same k order: the same bits. ----
    constexpr int PH_IN = 2 * (TH - 1) + KS, PW_IN = 2 * (TW - 1) + KS;      // 19 x 39
    constexpr int NPIX = PH_IN * PW_IN, ZERO_PIX = NPIX;                      // + one all-zero pixel for the taps beyond 49
    constexpr int PP_PLANE = ((NPIX + 1) * 8 + 15) / 16 * 16;
    unsigned char* PP = smem;                                                 // [2 planes][NPIX + 1][4 fp16]
    unsigned char* Ws = smem + 2 * PP_PLANE;                                  // [2 planes][64 rows x ROW_BYTES]
    {
        const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
        for (int t = tid; t <= NPIX; t += 256) {
            const int py = t / PW_IN, px = t - py * PW_IN;
            const int ih = ih0 + py, iw = iw0 + px;
            unsigned off = (unsigned)(((b * p.H + ih) * p.W + iw) * 16);
            if (t == ZERO_PIX || !(((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W))) off = INVALID;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
            u32x2 p0, p1;
            split4(v, p0, p1);
            *reinterpret_cast<u32x2*>(PP + t * 8) = p0;
            *reinterpret_cast<u32x2*>(PP + PP_PLANE + t * 8) = p1;
        }
    }
    const int wq = tid & 3;
    const unsigned w_off = (unsigned)((tid >> 2) * p.ldw + wq * 8) * 2u;     // 64 rows x four 16-byte chunks: one unit per thread
    const unsigned w_plane_bytes = (unsigned)(p.w_plane_stride * 2);
    u32x4 w_reg[2];
    auto load_W = [&](int kt) {
        const unsigned koff = (unsigned)(kt * BK) * 2u;      // planes are zero-padded to ldw (a multiple of 32)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            w_reg[pl] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(w_off + koff + pl * w_plane_bytes), 0, 0);
    };
    auto store_W = [&]() {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            *reinterpret_cast<u32x4*>(Ws + pl * W_PLANE + (tid >> 2) * ROW_BYTES + wq * 16) = w_reg[pl];
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    // patch pixel under tap (0, 0) of this lane's two tile rows (rows beyond the 7 x 17 patch compute on pixel 0: never used)
    int rb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wr * 64 + i * 32 + fr;
        const int ty = row / TW, tx = row - ty * TW;
        rb[i] = row < TH * TW ? (2 * ty) * PW_IN + 2 * tx : 0;
    }
    const unsigned char* w_base = Ws + (wc * 32 + fr) * ROW_BYTES + fh * 16;
    auto compute = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            // k-step 2 kt + ks: this half-wave's eight k values = taps t0, t0 + 1 (four channels each)
            const int t0 = (2 * kt + ks) * 4 + 2 * fh;
            int po[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tap = t0 + j, kh = tap / KS, kw = tap - kh * KS;
                po[j] = tap < TAPS ? kh * PW_IN + kw : -1;
            }
            half8 af[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(PP + pl * PP_PLANE + (po[0] < 0 ? ZERO_PIX : rb[i] + po[0]) * 8);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(PP + pl * PP_PLANE + (po[1] < 0 ? ZERO_PIX : rb[i] + po[1]) * 8);
                    af[pl][i] = __builtin_bit_cast(half8, (u32x4{lo[0], lo[1], hi[0], hi[1]}));
                }
            const half8 b0 = *reinterpret_cast<const half8*>(w_base + ks * 32);
            const half8 b1 = *reinterpret_cast<const half8*>(w_base + W_PLANE + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 c = acc[i];                           // smallest terms first (the tile kernel's order)
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], b0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], b1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], b0, c, 0, 0, 0);
                acc[i] = c;
            }
        }
    };

    constexpr int NK = (TAPS * 4 + BK - 1) / BK;             // 7 k-tiles
    load_W(0);
    store_W();
    __syncthreads();                                         // the patch and the first weight tile are staged
    for (int kt = 0; kt < NK; ++kt) {
        if (kt + 1 < NK) load_W(kt + 1);
        compute(kt);
        __syncthreads();
        if (kt + 1 < NK) {
            store_W();
            __syncthreads();
        }
    }

    // ---- epilogue: y = relu(acc * scale + shift) into the staged patch [128 rows][64 channels], then the 3x3 / 2 maxima ----
    float* patch = reinterpret_cast<float*>(smem);           // (the k-loop's last barrier has passed: its LDS is free)
    int bad = 0;
    {
        const int n = wc * 32 + fr;                          // this lane's channel
        float sc = p.scale ? p.scale[n] : 1.f;
        sc = sc * p.wscale[n];                               // exact: a power of two
        const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = acc[i][r] * sc + sh;
                bad |= !(fabsf(v) <= 3.4e38f);               // BEFORE the ReLU: max(NaN, 0) = 0 would hide an Inf - Inf
                v = fmaxf(v, 0.f);
                patch[row * PITCH + n] = v;
            }
    }
    __syncthreads();
    for (int t = tid; t < PTH * PTW * (BN / 4); t += 256) {
        const int q = t & 15, pp = t >> 4;
        const int py = pp / PTW, px = pp - py * PTW;
        const int ph = ph0 + py, pw = pw0 + px;
        if (ph >= p.PH || pw >= p.PW) continue;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ty = 2 * py + dy;
            if ((unsigned)(oh0 + ty) >= (unsigned)p.OH) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int tx = 2 * px + dx;
                if ((unsigned)(ow0 + tx) >= (unsigned)p.OW) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + (ty * TW + tx) * PITCH + 4 * q);
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        *reinterpret_cast<f32x4*>(p.Y + (((size_t)b * p.PH + ph) * p.PW + pw) * BN + 4 * q) = m;
    }
    if (bad && p.flag) atomicOr(p.flag, 1);                  // Inf / NaN: an operand left fp16's range (or came in bad)
}

}  // namespace

extern "C" int gom_stem_conv_pool_f32(const float* X, const void* Wplanes, long w_plane_stride, int ldw, const float* wscale,
                                      const float* scale, const float* shift, float* Y, int B, int H, int Wd, int* flag,
                                      void* stream) {
    GOM_CHECK_ARG(X && Wplanes && wscale && Y && B > 0 && H > 0 && Wd > 0);
    GOM_CHECK_ARG(ldw >= TAPS * 4 && (ldw % 32) == 0 && (w_plane_stride % 8) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)Wplanes % 16) == 0 && ((uintptr_t)Y % 16) == 0);
    GOM_CHECK_ARG((long)B * H * Wd * 4 < (1L << 29));        // 32-bit byte offsets with a 2 GiB range check
    StemArgs a{};
    a.X = X; a.Wp = (const unsigned short*)Wplanes; a.wscale = wscale; a.scale = scale; a.shift = shift; a.Y = Y; a.flag = flag;
    a.w_plane_stride = w_plane_stride; a.ldw = ldw; a.H = H; a.W = Wd;
    a.OH = (H + 6 - KS) / 2 + 1; a.OW = (Wd + 6 - KS) / 2 + 1;
    GOM_CHECK_ARG(a.OH > 0 && a.OW > 0);
    a.PH = (a.OH + 2 - 3) / 2 + 1; a.PW = (a.OW + 2 - 3) / 2 + 1;
    a.tiles_h = cdiv(a.PH, PTH); a.tiles_w = cdiv(a.PW, PTW);
    const long tiles = (long)B * a.tiles_h * a.tiles_w;
    GOM_CHECK_ARG(tiles < (1L << 31));
    hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)tiles), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return gom_launch_status();
}
