// Tail of a ResNet bottleneck block fused with the head of the next one for the WIDE stage (res4: 256 -> 1024 -> 256), two waves per SIMD
// (f16x3 split, fp32-class accuracy):
//
//   X  = relu( bn3(conv3_1x1(A)) + R )            A [M, 256] = the block's conv2 output, R [M, 1024] = its input
//   Y1 = relu( bn1'(conv1'_1x1(X)) )              the NEXT block's first convolution, 256 output channels
//
// (Detectron2 BottleneckBlock, STRIDE_IN_1X1 = False, FrozenBN folded: gom_lstmatcher.py:42-61 builds it)
// bneck_fused.hip serves res2 / res3; at res4's widths its one-workgroup-per-CU form measured no better than the two tile-kernel
// launches (285-328 vs 263-285 us at 56 448 pixels: the chain R load -> product -> X store -> product runs with one wave per SIMD
// and nothing to hide its latencies, 3 x its MFMA time).  Here a wave owns 16 pixels -- as v_mfma_f32_16x16x32_f16 operand fragments
// in 64 VGPRs, the 256 Y1 channels in 64 more: <= 256 registers -- and a workgroup is EIGHT waves = 128 pixels, two waves per SIMD
// sharing ONE weight ring: one wave's residual exchange / epilogue / X stores run under the other's MFMAs, and the weights cross
// L2 -> LDS once per 128 pixels.  (First built as two 4-wave workgroups of 64 pixels per CU with a half-stage ring each: 261-273 us
// against 292-326 for the two launches -- every wave then issues sixteen 1 KB LDS-DMA pieces per 96 MFMAs, 100-180 cycles each,
// and that issue, not the matrix pipe or the LDS, bounded the loop.)
//
// Per chunk of 32 X-channels (the fused FFN kernel's chunk, ffn_fused.hip, with the hidden activation also leaving the chip):
//     H^T[32 x 16 px]   = W3c . A^T                     48 MFMAs; weight fragments from LDS, the pixels' fragments in registers
//     v = relu(H^T * scale + shift + R^T)               R in / X out as WHOLE 128-byte lines through a wave-private LDS tile
//     Y1^T[256 x 16 px] += W1'[:, chunk] . v^T          48 MFMAs; v straight from the accumulator registers after the fp16 split
// The weight image IS gom_ffn_fused_image's (W1 := conv3's [1024, 256], W2 := conv1's [256, 1024], the 1 / row scale x BN scale and
// the BN shift in the stage's last fragment), streamed through a two-stage ring by MUBUF LDS-DMA, eight or nine pieces per wave and
// chunk, all issued inside the first product.  Every weight fragment serves 16 pixels (3 MFMAs per KB of LDS reads): the loop is
// bound by the LDS at ~75 % of the matrix pipe.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int CH = 32;                                   // X channels per chunk = one 128-byte line per pixel
constexpr int FRAG = 1024;
constexpr int WAVES = 8, BM = 16 * WAVES;
constexpr int XT_ROW = 36;                               // floats per pixel row of a wave's transpose tile (32 + 4: conflict-free b128)
constexpr int XT_BYTES = 16 * XT_ROW * 4;
// K1 = conv3's inputs (256: res4; 128: the res3 -> res4 transition), MP = conv1' outputs (256)
template <int K1, int MP>
struct Cfg2 {
    static constexpr int WA_FRAGS = (K1 / 32) * (CH / 16) * 2;   // first product: k-steps x channel groups x planes
    static constexpr int WB_FRAGS = (MP / 16) * 2;               // second product: output groups x planes
    static constexpr int STAGE_FRAGS = WA_FRAGS + WB_FRAGS + 1;  // (K1 = MP = 256: gom_ffn_fused_image's stage, 65 fragments)
    static constexpr int STAGE_BYTES = STAGE_FRAGS * FRAG;
    static constexpr int RING_BYTES = 2 * STAGE_BYTES;
    static constexpr int SCRATCH_BYTES = WAVES * 16 * 128 * 4;   // the prologue's rows -> fragments scratch (64 KB)
    static constexpr int LDS_BYTES = (RING_BYTES > STAGE_BYTES + SCRATCH_BYTES ? RING_BYTES : STAGE_BYTES + SCRATCH_BYTES) + WAVES * XT_BYTES;
    static constexpr int XT_OFF = LDS_BYTES - WAVES * XT_BYTES;
};

struct B2Args {
    const float* A;
    const unsigned char* img;
    const float* R;
    const float* sc1;                                        // [MP] folded scale (BN scale x 1 / weight row scale) and shift of conv1'
    const float* sh1;
    float* X;
    float* Y1;
    int* flag;
    int lda, ldr, ldx, ldy, M, chunks;
};

__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned byte_offset, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)byte_offset, 0, 0, 0);
}

template <int K1, int MP>
__global__ __launch_bounds__(512, 1) void bneck2_kernel(const B2Args p) {
    using C = Cfg2<K1, MP>;
    constexpr int WA_FRAGS = C::WA_FRAGS, WB_FRAGS = C::WB_FRAGS, STAGE_FRAGS = C::STAGE_FRAGS, STAGE_BYTES = C::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 15, fg = lane >> 4;
    const long row0 = (long)blockIdx.x * BM + wave * 16;
    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, p.chunks * STAGE_BYTES, 0x00020000);
    float* xt = reinterpret_cast<float*>(smem + C::XT_OFF + wave * XT_BYTES);

    // R in and X out move as whole 128-byte lines: lane l handles 16-byte piece (l & 7) of pixels (l >> 3) + 8 i of the wave's 16
    long crow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long mm = row0 + (lane >> 3) + 8 * i;
        crow[i] = mm < p.M ? mm : p.M - 1;                   // tail pixels recompute (and re-store) the last one: same bits
    }
    const int cpc = (lane & 7) * 4;
    const long myrow = row0 + fn < p.M ? row0 + fn : p.M - 1;

    // ---- this wave's 16 pixels of A as B-operand fragments (lane (n, kg) <- A[px n][32 s + 8 kg .. + 7]): whole-line loads + a layout
    // change in a wave-private 8 KB of slot B (free until the first chunk's second half is requested) ----
    float amax = 0.f, chk = 0.f;
    half8 xf[2][K1 / 32];
    f32x4 rv[2];                                             // the chunk's residual piece in the coalesced layout
    {
        float* scratch = reinterpret_cast<float*>(smem + STAGE_BYTES) + wave * (16 * 128);   // the ring's second slot: free until chunk 0 requests stage 1
        const int pc = lane & 31, r0 = lane >> 5;
#pragma unroll
        for (int part = 0; part < K1 / 128; ++part) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                long m = row0 + r0 + 2 * i;
                if (m > p.M - 1) m = p.M - 1;
                v[i] = *reinterpret_cast<const f32x4*>(p.A + (size_t)m * p.lda + part * 128 + pc * 4);
            }
            if (part == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) rv[i] = *reinterpret_cast<const f32x4*>(p.R + (size_t)crow[i] * p.ldr + cpc);
                __builtin_amdgcn_sched_barrier(0);
                for (int f = wave; f < STAGE_FRAGS; f += WAVES) dma_fragment(rs_img, f * FRAG + lane * 16, smem + f * FRAG);   // stage 0
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + 2 * i;
                *reinterpret_cast<f32x4*>(scratch + r * 128 + ((pc ^ (r & 31)) << 2)) = v[i];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int p0 = 8 * s_ + 2 * fg;
                const f32x4 a = *reinterpret_cast<const f32x4*>(scratch + fn * 128 + ((p0 ^ (fn & 31)) << 2));
                const f32x4 b = *reinterpret_cast<const f32x4*>(scratch + fn * 128 + (((p0 + 1) ^ (fn & 31)) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(a[e]), fabsf(b[e])));
                gom_split8_f16(a, b, xf[0][4 * part + s_], xf[1][4 * part + s_]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        asm volatile("" : "+v"(amax));
    }

    f32x4 acc2[MP / 16];
#pragma unroll
    for (int t = 0; t < MP / 16; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr unsigned OOB = 0x7FFF0000u;
#define B2_LOAD(dst, g)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
        dst[i_] = *reinterpret_cast<const half8*>(base + ((g) * 8 + i_) * FRAG);
#define B2_DMA(i)
#define B2_PIN()                                          \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    for (int c = 0; c < p.chunks; ++c) {
        const int st = c & 1;
        const bool more = c + 1 < p.chunks;
        // the next stage's 65 fragments: 64 + wave .. by wave 0 here, the others -- eight per wave -- inside the first product
        if (more && wave == 0)
            dma_fragment(rs_img, (unsigned)(c + 1) * STAGE_BYTES + (WA_FRAGS + WB_FRAGS) * FRAG + lane * 16,
                         smem + (st ^ 1) * STAGE_BYTES + (WA_FRAGS + WB_FRAGS) * FRAG);
        const unsigned nsrc = more ? (unsigned)(c + 1) * STAGE_BYTES + wave * FRAG + lane * 16 : OOB;
        unsigned char* ndst = smem + (st ^ 1) * STAGE_BYTES + wave * FRAG;
#undef B2_DMA
#define B2_DMA(i) dma_fragment(rs_img, nsrc + (i) * WAVES * FRAG, ndst + (i) * WAVES * FRAG);
        // ================= first product: H^T chunk = W3c . A^T =================
        f32x4 acc1[2];
        f32x4 rn[2];
        {
            const unsigned char* base = smem + st * STAGE_BYTES + lane * 16;
            // the next chunk's residual piece: the oldest vector-memory operations of the chunk
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (more) rn[i] = *reinterpret_cast<const f32x4*>(p.R + (size_t)crow[i] * p.ldr + CH * (c + 1) + cpc);
            }
            half8 fa[8], fb[8];
            acc1[0] = acc1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            // fragment 4 i + 2 Hh + p of a group = plane p of channel group Hh at the group's k-step i
#define B2_GEMM1(src, g)                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                        \
        const int s_ = (g) * 2 + i_;                                                                          \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc1[h_] = mfma16(src[4 * i_ + 2 * h_ + 1], xf[0][s_], acc1[h_]);  \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc1[h_] = mfma16(src[4 * i_ + 2 * h_], xf[1][s_], acc1[h_]);      \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc1[h_] = mfma16(src[4 * i_ + 2 * h_], xf[0][s_], acc1[h_]);      \
    }
            // (DMA pieces of this wave: fragments wave, wave + 8, ... below STAGE_FRAGS - 1; the last fragment went out above)
#define B2_DMAX(i) if ((i) * WAVES < WA_FRAGS + WB_FRAGS) B2_DMA(i)
            B2_LOAD(fa, 0)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            if constexpr (K1 == 256) {
                B2_LOAD(fb, 1) B2_GEMM1(fa, 0) B2_DMAX(0) B2_DMAX(1) B2_PIN()
                B2_LOAD(fa, 2) B2_GEMM1(fb, 1) B2_DMAX(2) B2_DMAX(3) B2_PIN()
                B2_LOAD(fb, 3) B2_GEMM1(fa, 2) B2_DMAX(4) B2_DMAX(5) B2_PIN()
                B2_GEMM1(fb, 3) B2_DMAX(6) B2_DMAX(7)
            } else {
                static_assert(K1 == 128 || K1 == 256, "K1");
                B2_LOAD(fb, 1) B2_GEMM1(fa, 0) B2_DMAX(0) B2_DMAX(1) B2_DMAX(2) B2_PIN()
                B2_GEMM1(fb, 1) B2_DMAX(3) B2_DMAX(4) B2_DMAX(5)
            }
#undef B2_DMAX
#undef B2_GEMM1
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- this chunk's residual piece: coalesced layout -> the wave's tile -> accumulator layout ----
        const float* aux = reinterpret_cast<const float*>(smem + st * STAGE_BYTES + (WA_FRAGS + WB_FRAGS) * FRAG);
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(xt + ((lane >> 3) + 8 * i) * XT_ROW + cpc) = rv[i];
        __builtin_amdgcn_wave_barrier();
        f32x4 ra[2];
#pragma unroll
        for (int h_ = 0; h_ < 2; ++h_) ra[h_] = *reinterpret_cast<const f32x4*>(xt + fn * XT_ROW + 16 * h_ + 4 * fg);
        __builtin_amdgcn_wave_barrier();
        // ---- X chunk = relu(acc * scale + shift + R): stored as whole lines, and split into the second product's B fragment ----
        half8 hf[2];
        {
            f32x4 v[2];
#pragma unroll
            for (int h_ = 0; h_ < 2; ++h_) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + 16 * h_ + 4 * fg);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(aux + CH + 16 * h_ + 4 * fg);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc1[h_][e] * sc[e] + sh[e] + ra[h_][e];     // the tile kernel's epilogue arithmetic
                    chk = fmaf(t, 0.f, chk);                                     // in front of the ReLU: NaN / Inf -> NaN
                    v[h_][e] = fmaxf(t, 0.f);
                    amax = fmaxf(amax, v[h_][e]);
                }
                *reinterpret_cast<f32x4*>(xt + fn * XT_ROW + 16 * h_ + 4 * fg) = v[h_];
            }
            gom_split8_f16(v[0], v[1], hf[0], hf[1]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(xt + ((lane >> 3) + 8 * i) * XT_ROW + cpc);
            *reinterpret_cast<f32x4*>(p.X + (size_t)crow[i] * p.ldx + CH * c + cpc) = o;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" : "+v"(amax), "+v"(chk));
        __builtin_amdgcn_sched_barrier(0);
        // ================= second product: Y1^T += W1'[:, chunk] . X^T =================
        {
            const unsigned char* base = smem + st * STAGE_BYTES + WA_FRAGS * FRAG + lane * 16;
            half8 fa[8], fb[8];
            // fragment 2 i + p of group g = plane p of output group 4 g + i
#define B2_GEMM2(src, g)                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int t_ = (g) * 4 + i_;                                                                          \
        acc2[t_] = mfma16(src[2 * i_ + 1], hf[0], acc2[t_]);                                                  \
        acc2[t_] = mfma16(src[2 * i_], hf[1], acc2[t_]);                                                      \
        acc2[t_] = mfma16(src[2 * i_], hf[0], acc2[t_]);                                                      \
    }
#define B2_PIN0()                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            static_assert(MP == 256, "MP");
            B2_LOAD(fa, 0)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            B2_LOAD(fb, 1) B2_GEMM2(fa, 0) B2_PIN0()
            B2_LOAD(fa, 2) B2_GEMM2(fb, 1) B2_PIN0()
            B2_LOAD(fb, 3) B2_GEMM2(fa, 2) B2_PIN0()
            B2_GEMM2(fb, 3)
#undef B2_GEMM2
#undef B2_PIN0
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) rv[i] = rn[i];
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // everything but this chunk's two X stores has landed
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    }
#undef B2_LOAD
#undef B2_DMA
#undef B2_PIN

    // ---- Y1 = relu(acc2 * scale + shift): lane = pixel, registers = channels 16 t + 4 fg .. + 3 ----
    float* yrow = p.Y1 + (size_t)myrow * p.ldy + 4 * fg;
#pragma unroll
    for (int t = 0; t < MP / 16; ++t) {
        const int col = 16 * t + 4 * fg;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.sc1 + col);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.sh1 + col);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float tv = acc2[t][e] * sc[e] + sh[e];
            chk = fmaf(tv, 0.f, chk);
            o[e] = fmaxf(tv, 0.f);
        }
        *reinterpret_cast<f32x4*>(yrow + 16 * t) = o;
    }
    if ((!(amax <= 65504.f) || !(chk == 0.f)) && p.flag) atomicOr(p.flag, 1);
}

// Fragment-linear weight image, per chunk c of 32 X-channels; element j of lane l = (m, kg) = (l & 15, l >> 4):
//   f = 4 s + 2 Hh + p       (s < K1 / 32, Hh < 2)  : plane p of W3s[32 c + 16 Hh + m][32 s + 8 kg + j]
//   f = WA + 2 t + p         (t < MP / 16)          : plane p of W1s[16 t + m][32 c + 16 (j >> 2) + 4 kg + (j & 3)]
//   f = WA + WB                                     : floats 0..31 = BN scale x 1 / row scale of W3s, 32..63 = BN shift of the chunk
__global__ __launch_bounds__(256) void bneck2_image_kernel(const unsigned short* __restrict__ p3, long ps3, int ld3,
                                                           const float* __restrict__ inv3, const float* __restrict__ scale3,
                                                           const float* __restrict__ shift3, const unsigned short* __restrict__ p1,
                                                           long ps1, int ld1, int k1, int c4, int mp, unsigned short* __restrict__ img) {
    const int wa = (k1 / 32) * 4, wb = (mp / 16) * 2, sf = wa + wb + 1;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)(c4 / CH) * sf * 512;
    if (i >= total) return;
    const int e = (int)(i % 512), f = (int)((i / 512) % sf), c = (int)(i / (512L * sf));
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    if (f < wa) {
        const int s_ = f >> 2, hh = (f >> 1) & 1, pl = f & 1;
        img[i] = p3[pl * ps3 + (size_t)(CH * c + 16 * hh + m) * ld3 + 32 * s_ + 8 * kg + j];
    } else if (f < wa + wb) {
        const int id = f - wa, t = id >> 1, pl = id & 1;
        img[i] = p1[pl * ps1 + (size_t)(16 * t + m) * ld1 + CH * c + 16 * (j >> 2) + 4 * kg + (j & 3)];
    } else {
        const int fi = e >> 1;
        float v = 0.f;
        if (fi < CH) v = inv3[CH * c + fi] * (scale3 ? scale3[CH * c + fi] : 1.f);   // exact: the row scale is a power of two
        else if (fi < 2 * CH) v = shift3 ? shift3[CH * c + fi - CH] : 0.f;
        const unsigned bits = __builtin_bit_cast(unsigned, v);
        img[i] = (unsigned short)((e & 1) ? (bits >> 16) : (bits & 0xffffu));
    }
}

bool served2(int k1, int c4, int mp) { return c4 == 4 * k1 && mp == 256 && (k1 == 256 || k1 == 128); }

template <int K1, int MP>
int launch2(const B2Args& a, hipStream_t s) {
    using C = Cfg2<K1, MP>;
    auto kern = bneck2_kernel<K1, MP>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(a.M, BM)), dim3(64 * WAVES), C::LDS_BYTES, s, a);
    return gom_launch_status();
}

}  // namespace

extern "C" long gom_bneck2_image_bytes(int k1, int c4, int mp) {
    if (!served2(k1, c4, mp)) return -1;
    return (long)(c4 / CH) * ((k1 / 32) * 4 + (mp / 16) * 2 + 1) * FRAG;
}

extern "C" int gom_bneck2_image(const void* w3_planes, long w3_plane_stride, int ld3, const float* w3_inv_scale, const float* scale3,
                                const float* shift3, const void* w1_planes, long w1_plane_stride, int ld1, int k1, int c4, int mp,
                                void* image, long image_bytes, void* stream) {
    GOM_CHECK_ARG(w3_planes && w3_inv_scale && w1_planes && image && served2(k1, c4, mp) && ld3 >= k1 && ld1 >= c4);
    GOM_CHECK_ARG(image_bytes >= gom_bneck2_image_bytes(k1, c4, mp));
    const long total = (long)(c4 / CH) * ((k1 / 32) * 4 + (mp / 16) * 2 + 1) * 512;
    hipLaunchKernelGGL(bneck2_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w3_planes, w3_plane_stride, ld3, w3_inv_scale, scale3, shift3,
                       (const unsigned short*)w1_planes, w1_plane_stride, ld1, k1, c4, mp, (unsigned short*)image);
    return gom_launch_status();
}

/* image: gom_bneck2_image */
extern "C" int gom_bneck2_f32(const float* A, int lda, const void* image, const float* R, int ldr, const float* scale1,
                              const float* shift1, float* X, int ldx, float* Y1, int ldy, int M, int k1, int c4, int mp, int* flag,
                              void* stream) {
    GOM_CHECK_ARG(A && image && R && scale1 && shift1 && X && Y1 && M >= 0 && served2(k1, c4, mp));
    GOM_CHECK_ARG(lda >= k1 && ldr >= c4 && ldx >= c4 && ldy >= mp && (lda % 4) == 0 && (ldr % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)R % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y1 % 16) == 0 &&
                  ((uintptr_t)image % 16) == 0 && ((uintptr_t)scale1 % 16) == 0 && ((uintptr_t)shift1 % 16) == 0);
    if (M == 0) return GOM_OK;
    B2Args a{};
    a.A = A; a.img = (const unsigned char*)image; a.R = R; a.sc1 = scale1; a.sh1 = shift1; a.X = X; a.Y1 = Y1; a.flag = flag;
    a.lda = lda; a.ldr = ldr; a.ldx = ldx; a.ldy = ldy; a.M = M; a.chunks = c4 / CH;
    if (k1 == 256) return launch2<256, 256>(a, (hipStream_t)stream);
    return launch2<128, 256>(a, (hipStream_t)stream);
}
