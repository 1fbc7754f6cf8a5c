// The row-local TAIL of a DeepSolo composite decoder layer as ONE launch (f16x3 split, fp32-class accuracy):
//
//   tgt    = norm3(tgt + linear2(relu(linear1(tgt))))                          the layer's FFN block       (32 weight chunks)
//   h      = relu(W2c relu(W1c tgt + b1c) + b2c)                               ctrl_point_coord[lid], hidden layers (8 chunks)
//   ref'   = sigmoid(W3c h + b3c + inverse_sigmoid(ref))                       its 256 -> 2 layer + the reference refinement
//   qpos'  = W2r relu(W1r sine_embed(ref') + b1r) + b2r                        the NEXT layer's ref_point_head  (8 chunks)
//
// (/root/reference/third_party/adet/layers/deformable_transformer.py:352-354,368-369 `forward_ffn` + norm3, :484-488 the
// refinement, :470-473 + adet/modeling/model/utils.py:24-37 the next layer's query position).  Until round 5 these were four
// launches of M = frames x queries x points = 20 000 rows each -- fused FFN, two-layer perceptron, `ref_update`, two-layer
// perceptron: 89 + 33 + ~8 + 33 us per layer, 157 one-per-CU workgroups each, every launch paying its own row load / fp16 split
// prologue (14k cycles), LDS-staged epilogue (13k) and dispatch gap (~5 us) around 8 chunks of work (35k cycles).  Everything
// here is row-local, so the rows stay on their lanes from the first load to the last store:
//
//   * the chunk pipeline is ffn_fused.hip's, unchanged (one wave per SIMD, 32 rows per wave as B-operand fragments in 128
//     VGPRs, v_mfma_f32_16x16x32_f16, weights as a fragment-linear image through a two-stage LDS ring by MUBUF LDS-DMA) and runs
//     over ONE concatenated image of 32 + 8 + 8 chunks, so the ring never drains between the blocks: the first stage of the next
//     block lands while the previous block's epilogue runs;
//   * the epilogues stay in REGISTERS.  A product comes out as Y^T (row = lane n, lane group g holds features 16 t + 4 g + i):
//     residual, bias and scale are loaded in that layout (64-byte pieces of 16 rows per instruction), the LayerNorm statistics
//     are in-lane sums over 64 values + two cross-lane steps, and -- ffn_fused.hip's own trick between its two products -- the
//     finished values ARE the next block's B operand after an fp16 split of registers (2 s, 2 s + 1): k-slot j of lane group g at
//     k-step s <-> feature 32 s + 16 (j >> 2) + 4 g + (j & 3), an order baked into the following block's W1 image (`perm` form
//     of the image kernel).  No LDS staging, no row reload, no layout change;
//   * the 256 -> 2 layer is 2 x 64 in-lane FMAs + two cross-lane steps in exact fp32; sigmoid / inverse_sigmoid as
//     elementwise.hip; the sine embedding is evaluated directly in operand order (each (row, feature) by exactly one lane: 64
//     sincosf per lane and row group, the pair (2 k, 2 k + 1) sharing its angle).
// Per row the arithmetic does not depend on what shares the launch (batch invariance); against the four-launch path the
// LayerNorm sums and the second / third block's k order differ in the last bits (tests hold both to the same fp64 tolerance).
#include "common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int D = 256;                                   // model width
constexpr int CH = 32;                                   // hidden units per weight chunk
constexpr int FRAG = 1024;
constexpr int W1_FRAGS = (D / 32) * (CH / 16) * 2;
constexpr int W2_FRAGS = (D / 16) * 2;
constexpr int STAGE_FRAGS = W1_FRAGS + W2_FRAGS + 1;     // ffn_fused.hip's stage: 65 fragments
constexpr int STAGE_BYTES = STAGE_FRAGS * FRAG;
constexpr int BM = 128;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
constexpr int MLP_CHUNKS = D / CH;                       // a 256 -> 256 -> 256 perceptron: 8 chunks
constexpr int LIN_STAGES = 4;                            // a 256 -> 256 linear layer: 4 stages of two 32-wide k-steps x 16 output groups

struct TailArgs {
    const float* X;                                          // tgt behind norm_cross [M, 256]
    const unsigned char* img;                                // FFN | ctrl_point_coord hidden layers | ref_point_head, concatenated
    const float *s2, *b2, *gamma, *beta;                     // FFN: 1 / row scale of W2, bias, norm3
    const float *c_s2, *c_b2;                                // coordinate MLP, second hidden layer
    const float *W3, *b3;                                    // its last layer [2, 256], [2]
    const float* ref;                                        // reference points [M, 2]
    const float* dim_t;                                      // [128]
    const float *q_s2, *q_b2;                                // ref_point_head, second layer
    float *Y, *new_ref, *QP;                                 // tgt out, refined references, next layer's query position (or null)
    int* flag;
    float eps;
    int ldx, ldy, ldq, M, ffn_chunks;
    // WITH_PROJ: X = the cross-attention's sampled rows, R = tgt in front of the block (the residual of norm_cross)
    const float *R, *p_s, *p_b, *p_gamma, *p_beta;
    float p_eps;
    int ldr;
};

__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned byte_offset, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)byte_offset, 0, 0, 0);
}

__device__ __forceinline__ float inv_sigmoid(float x) {   // adet/utils/misc.py:115-119, eps 1e-5 (as elementwise.hip)
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
    return logf(x1 / x2);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// sin and cos of an angle in [0, 2 pi] (the sine embedding's range: a reference point in [0, 1] times 2 pi over dim_t >= 1):
// quadrant by a two-term Cody-Waite reduction (exact with fma for q <= 4), then the cephes single-precision kernels on
// [-pi/4, pi/4] -- within 1 ulp of 1 of the correctly rounded values.  Not ocml's sinf / cosf: their large-argument path keeps
// the compiler from unrolling the embedding loop, and a dynamically indexed operand array goes to scratch.
__device__ __forceinline__ void sincos_0_2pi(float a, float& sn, float& cs) {
    const float q = rintf(a * 0.63661977236758134f);
    float r = fmaf(q, -1.57079637050628662109375f, a);
    r = fmaf(q, 4.37113900018624283e-8f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.f));
    const int qi = (int)q;
    const float s0 = (qi & 1) ? pc : ps, c0 = (qi & 1) ? ps : pc;
    sn = (qi & 2) ? -s0 : s0;
    cs = ((qi + 1) & 2) ? -c0 : c0;
}

// sum over the four lane groups (lanes n, n + 16, n + 32, n + 48), result in all of them
__device__ __forceinline__ float groups_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// (Measured, round 5: the FFN block ALONE on this kernel's register epilogue -- residual, LayerNorm and 16-byte stores in the
// accumulator layout instead of ffn_fused.hip's LDS-staged whole-row pass -- is SLOWER: 884 vs 811 us at M = 297 368, 83 vs 79 us at
// M = 20 000.  The register epilogue pays here only because it is what lets the blocks chain.)
template <bool WITH_QPOS, bool WITH_PROJ>
__global__ __launch_bounds__(256, 1) void dec_tail_kernel(const TailArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 15, fg = lane >> 4;
    const long tile0 = (long)blockIdx.x * BM;
    const long row0 = tile0 + wave * 32;
    constexpr int C0 = WITH_PROJ ? LIN_STAGES : 0;           // stages of the out_proj block in front of the FFN's
    const int total_chunks = C0 + p.ffn_chunks + MLP_CHUNKS + (WITH_QPOS ? MLP_CHUNKS : 0);

    unsigned pf[4];
    gom_prefetch_image(p.img, (unsigned)(total_chunks * STAGE_BYTES), tid, 256, pf);        // (common.h: a one-round launch, the image cold)
    const __amdgpu_buffer_rsrc_t rs_img =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, total_chunks * STAGE_BYTES, 0x00020000);
    auto dma_stage = [&](int c, int stage) {
        const unsigned src = (unsigned)c * STAGE_BYTES + lane * 16;
        unsigned char* dst = smem + stage * STAGE_BYTES;
        for (int f = wave; f < STAGE_FRAGS; f += 4) dma_fragment(rs_img, src + f * FRAG, dst + f * FRAG);
    };

    // this lane's two rows (row group r: row0 + 16 r + fn), clamped for loads; tail rows are never stored
    long mrow[2];
    bool live[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const long m = row0 + 16 * r + fn;
        live[r] = m < p.M;
        mrow[r] = live[r] ? m : p.M - 1;
    }

    float amax = 0.f, chk = 0.f;                             // range bookkeeping as dec_attn.hip: running |max| of what is split, NaN detector
    half8 xf[2][D / 16];
    {
        auto xrow = [&](int r) {
            long m = row0 + r;
            if (m > p.M - 1) m = p.M - 1;
            return p.X + (size_t)m * p.ldx;
        };
        gom_rows_to_fragments_t<128, false, true>(xrow, xrow, reinterpret_cast<float*>(smem + STAGE_BYTES) + wave * (32 * 128), lane, xf,
                                                  amax, [&]() { dma_stage(0, 0); });
    }

    f32x4 acc2[D / 16][2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gom_prefetch_done(pf);
    __syncthreads();

    float hmax = 0.f;
    constexpr unsigned OOB = 0x7FFF0000u;

    // ---- one weight chunk (global index c): H^T chunk = W1c . X^T, relu / scale / bias / split, Y^T += W2[:, c] . H^T ----
    // (ffn_fused.hip's loop body: two-deep fragment pipeline pinned with sched_group_barrier, the next stage's LDS-DMA one per
    //  six MFMAs)
#define DT_DMA(i) dma_fragment(rs_img, nsrc + (i) * 4 * FRAG, ndst + (i) * 4 * FRAG);
#define DT_LOAD(dst, g)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
        dst[i_] = *reinterpret_cast<const half8*>(base + ((g) * 8 + i_) * FRAG);
#define DT_PIN3()                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#define DT_PIN1()                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);   \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#define DT_PIN0()                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
#define DT_GEMM1(src, g)                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                        \
        const int s_ = (g) * 2 + i_;                                                                          \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                                  \
                acc1[h_][r_] = mfma16(src[4 * i_ + 2 * h_ + 1], xf[0][8 * r_ + s_], acc1[h_][r_]);            \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                                  \
                acc1[h_][r_] = mfma16(src[4 * i_ + 2 * h_], xf[1][8 * r_ + s_], acc1[h_][r_]);                \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                                  \
                acc1[h_][r_] = mfma16(src[4 * i_ + 2 * h_], xf[0][8 * r_ + s_], acc1[h_][r_]);                \
    }
#define DT_GEMM2(src, g)                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int t_ = (g) * 4 + i_;                                                                          \
        _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                                    \
            acc2[t_][r_] = mfma16(src[2 * i_ + 1], hf[0][r_], acc2[t_][r_]);                                  \
            acc2[t_][r_] = mfma16(src[2 * i_], hf[1][r_], acc2[t_][r_]);                                      \
            acc2[t_][r_] = mfma16(src[2 * i_], hf[0][r_], acc2[t_][r_]);                                      \
        }                                                                                                     \
    }
#define DT_CHUNK(c)                                                                                           \
    {                                                                                                         \
        const int st = (c) & 1;                                                                               \
        const bool more = (c) + 1 < total_chunks;                                                             \
        if (more && wave == 0)                                                                                \
            dma_fragment(rs_img, (unsigned)((c) + 1) * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG + lane * 16,  \
                         smem + (st ^ 1) * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG);                       \
        const unsigned nsrc = more ? (unsigned)((c) + 1) * STAGE_BYTES + wave * FRAG + lane * 16 : OOB;       \
        unsigned char* ndst = smem + (st ^ 1) * STAGE_BYTES + wave * FRAG;                                    \
        const unsigned char* base = smem + st * STAGE_BYTES + lane * 16;                                      \
        half8 fa[8], fb[8];                                                                                   \
        f32x4 acc1[2][2];                                                                                     \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) acc1[h_][r_] = f32x4{0.f, 0.f, 0.f, 0.f};        \
        DT_LOAD(fa, 0)                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                    \
        DT_LOAD(fb, 1) DT_GEMM1(fa, 0) DT_DMA(0) DT_DMA(1) DT_DMA(2) DT_PIN3()                                \
        DT_LOAD(fa, 2) DT_GEMM1(fb, 1) DT_DMA(3) DT_DMA(4) DT_DMA(5) DT_PIN3()                                \
        DT_LOAD(fb, 3) DT_GEMM1(fa, 2) DT_DMA(6) DT_DMA(7) DT_DMA(8) DT_PIN3()                                \
        DT_LOAD(fa, 4) DT_GEMM1(fb, 3) DT_DMA(9) DT_DMA(10) DT_DMA(11) DT_PIN3()                              \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        const float* aux = reinterpret_cast<const float*>(smem + st * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG);  \
        half8 hf[2][2];                                                                                       \
        {                                                                                                     \
            f32x4 sc[2], bi[2];                                                                               \
            _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                \
                sc[h_] = *reinterpret_cast<const f32x4*>(aux + 16 * h_ + 4 * fg);                             \
                bi[h_] = *reinterpret_cast<const f32x4*>(aux + CH + 16 * h_ + 4 * fg);                        \
            }                                                                                                 \
            _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                                \
                f32x4 v[2];                                                                                   \
                _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                              \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
                        v[h_][e] = fmaxf(fmaf(acc1[h_][r_][e], sc[h_][e], bi[h_][e]), 0.f);                   \
                        hmax = fmaxf(hmax, v[h_][e]);                                                         \
                    }                                                                                         \
                gom_split8_f16(v[0], v[1], hf[0][r_], hf[1][r_]);                                             \
            }                                                                                                 \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        DT_LOAD(fb, 5) DT_GEMM2(fa, 0) DT_DMA(12) DT_DMA(13) DT_DMA(14) DT_PIN3()                             \
        DT_LOAD(fa, 6) DT_GEMM2(fb, 1) DT_DMA(15) DT_PIN1()                                                   \
        DT_LOAD(fb, 7) DT_GEMM2(fa, 2) DT_PIN0()                                                              \
        DT_GEMM2(fb, 3)                                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        __syncthreads();                                                                                      \
    }
#define DT_ZERO_ACC2()                                                                                        \
    _Pragma("unroll") for (int t = 0; t < D / 16; ++t)                                                        \
        _Pragma("unroll") for (int r = 0; r < 2; ++r) acc2[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- one stage of a plain 256 -> 256 layer: Y^T += W[:, k-steps 2 c, 2 c + 1] . X^T; fragments [k-step u][output group t][plane] ----
#define DT_GEMM2X(src, g, ks)                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int t_ = ((g) & 3) * 4 + i_;                                                                    \
        _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_) {                                                    \
            acc2[t_][r_] = mfma16(src[2 * i_ + 1], xf[0][8 * r_ + (ks)], acc2[t_][r_]);                       \
            acc2[t_][r_] = mfma16(src[2 * i_], xf[1][8 * r_ + (ks)], acc2[t_][r_]);                           \
            acc2[t_][r_] = mfma16(src[2 * i_], xf[0][8 * r_ + (ks)], acc2[t_][r_]);                           \
        }                                                                                                     \
    }
#define DT_LIN(c)                                                                                             \
    {                                                                                                         \
        const int st = (c) & 1;                                                                               \
        if (wave == 0)                                                                                        \
            dma_fragment(rs_img, (unsigned)((c) + 1) * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG + lane * 16,  \
                         smem + (st ^ 1) * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG);                       \
        const unsigned nsrc = (unsigned)((c) + 1) * STAGE_BYTES + wave * FRAG + lane * 16;                    \
        unsigned char* ndst = smem + (st ^ 1) * STAGE_BYTES + wave * FRAG;                                    \
        const unsigned char* base = smem + st * STAGE_BYTES + lane * 16;                                      \
        half8 fa[8], fb[8];                                                                                   \
        DT_LOAD(fa, 0)                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                    \
        DT_LOAD(fb, 1) DT_GEMM2X(fa, 0, 2 * (c)) DT_DMA(0) DT_DMA(1) DT_DMA(2) DT_PIN3()                      \
        DT_LOAD(fa, 2) DT_GEMM2X(fb, 1, 2 * (c)) DT_DMA(3) DT_DMA(4) DT_DMA(5) DT_PIN3()                      \
        DT_LOAD(fb, 3) DT_GEMM2X(fa, 2, 2 * (c)) DT_DMA(6) DT_DMA(7) DT_DMA(8) DT_PIN3()                      \
        DT_LOAD(fa, 4) DT_GEMM2X(fb, 3, 2 * (c)) DT_DMA(9) DT_DMA(10) DT_DMA(11) DT_PIN3()                    \
        DT_LOAD(fb, 5) DT_GEMM2X(fa, 4, 2 * (c) + 1) DT_DMA(12) DT_DMA(13) DT_DMA(14) DT_PIN3()               \
        DT_LOAD(fa, 6) DT_GEMM2X(fb, 5, 2 * (c) + 1) DT_DMA(15) DT_PIN1()                                     \
        DT_LOAD(fb, 7) DT_GEMM2X(fa, 6, 2 * (c) + 1) DT_PIN0()                                                \
        DT_GEMM2X(fb, 7, 2 * (c) + 1)                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        __syncthreads();                                                                                      \
    }

    // ================================ block 0: out_proj of the cross attention + norm_cross ================================
    if constexpr (WITH_PROJ) {
        DT_ZERO_ACC2()
        DT_LIN(0) DT_LIN(1) DT_LIN(2) DT_LIN(3)
        float sum[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < D / 16; ++t) {
            const int col = 16 * t + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.p_s + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.p_b + col);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const f32x4 xr = *reinterpret_cast<const f32x4*>(p.R + (size_t)mrow[r] * p.ldr + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(acc2[t][r][e], sc[e], bi[e]) + xr[e];
                    acc2[t][r][e] = v;
                    sum[r] += v;
                }
            }
        }
        float mean[2], rstd[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) mean[r] = groups_sum(sum[r]) * (1.f / D);
        float sq[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < D / 16; ++t)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc2[t][r][e] -= mean[r];
                    sq[r] = fmaf(acc2[t][r][e], acc2[t][r][e], sq[r]);
                }
#pragma unroll
        for (int r = 0; r < 2; ++r) rstd[r] = rsqrtf(groups_sum(sq[r]) * (1.f / D) + p.p_eps);
#pragma unroll
        for (int s = 0; s < D / 32; ++s) {
            f32x4 o[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int t = 2 * s + hh, col = 16 * t + 4 * fg;
                const f32x4 ga = *reinterpret_cast<const f32x4*>(p.p_gamma + col);
                const f32x4 be = *reinterpret_cast<const f32x4*>(p.p_beta + col);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[hh][r][e] = acc2[t][r][e] * rstd[r] * ga[e] + be[e];
                        chk = fmaf(o[hh][r][e], 0.f, chk);
                        amax = fmaxf(amax, fabsf(o[hh][r][e]));
                    }
                    // tgt behind norm_cross is the FFN's residual: parked in Y (the same lane reads its own pieces back in the
                    // FFN's epilogue and then overwrites them with the block's result)
                    if (live[r]) *reinterpret_cast<f32x4*>(p.Y + (size_t)(row0 + 16 * r + fn) * p.ldy + col) = o[hh][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) gom_split8_f16(o[0][r], o[1][r], xf[0][8 * r + s], xf[1][8 * r + s]);
        }
        asm volatile("" : "+v"(amax), "+v"(chk));
        __builtin_amdgcn_sched_barrier(0);
    }

    // ================================ block 1: the FFN ================================
    DT_ZERO_ACC2()
    for (int c = C0; c < C0 + p.ffn_chunks; ++c) DT_CHUNK(c)

    // ---- epilogue 1 in registers: + bias, + residual, LayerNorm (two-pass), store, and the result as the next B operand ----
    {
        float sum[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < D / 16; ++t) {
            const int col = 16 * t + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s2 + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b2 + col);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                // (WITH_PROJ: the residual parked in Y by this lane above; a tail lane has parked nothing and must not read row
                //  M - 1 while its owner stores there -- ADVICE r5 -- so it takes zeros: its values are never stored)
                f32x4 xr = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (WITH_PROJ) {
                    if (live[r]) xr = *reinterpret_cast<const f32x4*>(p.Y + (size_t)mrow[r] * p.ldy + col);
                } else {
                    xr = *reinterpret_cast<const f32x4*>(p.X + (size_t)mrow[r] * p.ldx + col);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(acc2[t][r][e], sc[e], bi[e]) + xr[e];
                    acc2[t][r][e] = v;
                    sum[r] += v;
                }
            }
        }
        float mean[2], rstd[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) mean[r] = groups_sum(sum[r]) * (1.f / D);
        float sq[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < D / 16; ++t)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc2[t][r][e] -= mean[r];
                    sq[r] = fmaf(acc2[t][r][e], acc2[t][r][e], sq[r]);
                }
#pragma unroll
        for (int r = 0; r < 2; ++r) rstd[r] = rsqrtf(groups_sum(sq[r]) * (1.f / D) + p.eps);
#pragma unroll
        for (int s = 0; s < D / 32; ++s) {
            f32x4 o[2][2];                                       // [half of the k-step][row group]
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int t = 2 * s + hh, col = 16 * t + 4 * fg;
                const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + col);
                const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + col);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[hh][r][e] = acc2[t][r][e] * rstd[r] * ga[e] + be[e];
                        chk = fmaf(o[hh][r][e], 0.f, chk);
                        amax = fmaxf(amax, fabsf(o[hh][r][e]));
                    }
                    if (live[r]) *reinterpret_cast<f32x4*>(p.Y + (size_t)(row0 + 16 * r + fn) * p.ldy + col) = o[hh][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) gom_split8_f16(o[0][r], o[1][r], xf[0][8 * r + s], xf[1][8 * r + s]);
        }
        asm volatile("" : "+v"(amax), "+v"(chk));
    }
    __builtin_amdgcn_sched_barrier(0);

    // ================================ block 2: ctrl_point_coord's hidden layers ================================
    DT_ZERO_ACC2()
    for (int c = C0 + p.ffn_chunks; c < C0 + p.ffn_chunks + MLP_CHUNKS; ++c) DT_CHUNK(c)

    // ---- epilogue 2: relu(. / scale + bias), the 256 -> 2 layer in exact fp32, reference refinement ----
    float nref[2][2];                                        // [row group][x, y]
    {
        float dx[2] = {0.f, 0.f}, dy[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < D / 16; ++t) {
            const int col = 16 * t + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.c_s2 + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.c_b2 + col);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.W3 + col);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(p.W3 + D + col);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = fmaf(acc2[t][r][e], sc[e], bi[e]);
                    chk = fmaf(h, 0.f, chk);                     // (in front of the ReLU: max(NaN, 0) = 0 would hide it)
                    const float hr = fmaxf(h, 0.f);
                    dx[r] = fmaf(hr, w0[e], dx[r]);
                    dy[r] = fmaf(hr, w1[e], dy[r]);
                }
        }
        const float bx = p.b3[0], by = p.b3[1];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float ddx = groups_sum(dx[r]) + bx, ddy = groups_sum(dy[r]) + by;
            const float rx0 = p.ref[mrow[r] * 2], ry0 = p.ref[mrow[r] * 2 + 1];
            nref[r][0] = sigmoidf(ddx + inv_sigmoid(rx0));
            nref[r][1] = sigmoidf(ddy + inv_sigmoid(ry0));
            if (live[r] && fg == 0) *reinterpret_cast<f32x2*>(p.new_ref + (row0 + 16 * r + fn) * 2) = f32x2{nref[r][0], nref[r][1]};
        }
        asm volatile("" : "+v"(chk));
    }

    if constexpr (WITH_QPOS) {
        // ---- the next layer's point embedding (gen_point_pos_embed: channels [0, 128) <- x, [128, 256) <- y, sin on even, cos on
        //      odd channels of a pair that shares dim_t) directly as B-operand fragments in accumulator order ----
#pragma unroll
        for (int s = 0; s < D / 32; ++s) {
            f32x4 dt[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) dt[hh] = *reinterpret_cast<const f32x4*>(p.dim_t + ((32 * s + 16 * hh + 4 * fg) & 127));
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float e = nref[r][s >= 4 ? 1 : 0] * 6.283185307179586f;
                f32x4 o[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {                // channels (4 g + 2 k, 4 g + 2 k + 1) of the quad: one angle
                        const float a = e / dt[hh][2 * k];
                        float sn, cs;
                        sincos_0_2pi(a, sn, cs);
                        o[hh][2 * k] = sn;
                        o[hh][2 * k + 1] = cs;
                    }
                gom_split8_f16(o[0], o[1], xf[0][8 * r + s], xf[1][8 * r + s]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ================================ block 3: ref_point_head ================================
        DT_ZERO_ACC2()
        for (int c = C0 + p.ffn_chunks + MLP_CHUNKS; c < total_chunks; ++c) DT_CHUNK(c)

#pragma unroll
        for (int t = 0; t < D / 16; ++t) {
            const int col = 16 * t + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.q_s2 + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.q_b2 + col);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = fmaf(acc2[t][r][e], sc[e], bi[e]);
                    chk = fmaf(o[e], 0.f, chk);
                }
                if (live[r]) *reinterpret_cast<f32x4*>(p.QP + (size_t)(row0 + 16 * r + fn) * p.ldq + col) = o;
            }
        }
        asm volatile("" : "+v"(chk));
    }
#undef DT_DMA
#undef DT_LOAD
#undef DT_PIN3
#undef DT_PIN1
#undef DT_PIN0
#undef DT_GEMM1
#undef DT_GEMM2
#undef DT_CHUNK
#undef DT_LIN
#undef DT_GEMM2X
#undef DT_ZERO_ACC2
    // an operand left fp16's range, or a result is not finite (gemm_f16x3.hip contract; fmaxf drops a NaN, `chk` catches it)
    if ((!(amax <= 65504.f) || !(hmax <= 65504.f) || !(chk == 0.f)) && p.flag) atomicOr(p.flag, 1);
}

// Fragment-linear image of a plain 256 -> 256 layer for DT_LIN: stage c (0..3), fragment f = 32 u + 2 t + p (u = 0, 1; t = 0..15):
// element j of lane (m, kg) = plane p of Ws[16 t + m][32 (2 c + u) + 8 kg + j] (Ws = the row-scaled planes of gom_split_f16x2);
// fragment 64 is unused.  The input rows arrive in the standard operand layout (gom_rows_to_fragments_t, K32).
__global__ __launch_bounds__(256) void lin_image_kernel(const unsigned short* __restrict__ planes, long ps, int ld,
                                                        unsigned short* __restrict__ img) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)LIN_STAGES * STAGE_FRAGS * 512;
    if (i >= total) return;
    const int e = (int)(i % 512), f = (int)((i / 512) % STAGE_FRAGS), c = (int)(i / (512L * STAGE_FRAGS));
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    if (f >= 64) {
        img[i] = 0;
        return;
    }
    const int u = f >> 5, t = (f >> 1) & 15, pl = f & 1;
    img[i] = planes[pl * ps + (size_t)(16 * t + m) * ld + 32 * (2 * c + u) + 8 * kg + j];
}

}  // namespace

extern "C" long gom_dec_tail_image_bytes(int d_model, int d_hidden, int with_qpos) {
    if (d_model != D || d_hidden <= 0 || (d_hidden % CH) != 0) return -1;
    return (long)(d_hidden / CH + MLP_CHUNKS + (with_qpos ? MLP_CHUNKS : 0)) * STAGE_BYTES;
}

extern "C" long gom_dec_tail_lin_image_bytes(void) { return (long)LIN_STAGES * STAGE_BYTES; }

extern "C" int gom_dec_tail_lin_image(const void* w_planes, long w_plane_stride, int ld, void* image, long image_bytes, void* stream) {
    GOM_CHECK_ARG(w_planes && image && ld >= D && image_bytes >= (long)LIN_STAGES * STAGE_BYTES);
    const long total = (long)LIN_STAGES * STAGE_FRAGS * 512;
    hipLaunchKernelGGL(lin_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_planes, w_plane_stride, ld, (unsigned short*)image);
    return gom_launch_status();
}

static int dec_tail_launch(TailArgs& a, bool with_proj, void* stream) {
    const void* k[4] = {(const void*)dec_tail_kernel<false, false>, (const void*)dec_tail_kernel<true, false>,
                        (const void*)dec_tail_kernel<false, true>, (const void*)dec_tail_kernel<true, true>};
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    }
    const dim3 grid((unsigned)cdiv(a.M, BM)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (with_proj) {
        if (a.QP) hipLaunchKernelGGL((dec_tail_kernel<true, true>), grid, block, LDS_BYTES, s, a);
        else hipLaunchKernelGGL((dec_tail_kernel<false, true>), grid, block, LDS_BYTES, s, a);
    } else {
        if (a.QP) hipLaunchKernelGGL((dec_tail_kernel<true, false>), grid, block, LDS_BYTES, s, a);
        else hipLaunchKernelGGL((dec_tail_kernel<false, false>), grid, block, LDS_BYTES, s, a);
    }
    return gom_launch_status();
}

#define GOM_ALIGNED16(ptr) (((uintptr_t)(ptr) % 16) == 0)

extern "C" int gom_dec_tail_f32(const float* X, int ldx, const void* image, int d_hidden, const float* w2_inv_scale, const float* b2,
                                const float* gamma, const float* beta, float eps, const float* c_inv_scale, const float* c_b2,
                                const float* W3, const float* b3, const float* ref, const float* dim_t128,
                                const float* q_inv_scale, const float* q_b2, float* Y, int ldy, float* new_ref, float* qpos, int ldq,
                                int M, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && w2_inv_scale && b2 && gamma && beta && c_inv_scale && c_b2 && W3 && b3 && ref && Y && new_ref);
    GOM_CHECK_ARG(M >= 0 && d_hidden > 0 && (d_hidden % CH) == 0 && ldx >= D && ldy >= D && (ldx % 4) == 0 && (ldy % 4) == 0);
    GOM_CHECK_ARG(!qpos || (dim_t128 && q_inv_scale && q_b2 && ldq >= D && (ldq % 4) == 0 && GOM_ALIGNED16(qpos)));
    GOM_CHECK_ARG(GOM_ALIGNED16(X) && GOM_ALIGNED16(Y) && GOM_ALIGNED16(image) && ((uintptr_t)new_ref % 8) == 0);
    GOM_CHECK_ARG(GOM_ALIGNED16(w2_inv_scale) && GOM_ALIGNED16(b2) && GOM_ALIGNED16(gamma) && GOM_ALIGNED16(beta) &&
                  GOM_ALIGNED16(c_inv_scale) && GOM_ALIGNED16(c_b2) && GOM_ALIGNED16(W3));
    GOM_CHECK_ARG(!qpos || (GOM_ALIGNED16(dim_t128) && GOM_ALIGNED16(q_inv_scale) && GOM_ALIGNED16(q_b2)));
    if (M == 0) return GOM_OK;
    TailArgs a{};
    a.X = X; a.img = (const unsigned char*)image; a.s2 = w2_inv_scale; a.b2 = b2; a.gamma = gamma; a.beta = beta;
    a.c_s2 = c_inv_scale; a.c_b2 = c_b2; a.W3 = W3; a.b3 = b3; a.ref = ref; a.dim_t = dim_t128; a.q_s2 = q_inv_scale; a.q_b2 = q_b2;
    a.Y = Y; a.new_ref = new_ref; a.QP = qpos; a.flag = flag; a.eps = eps; a.ldx = ldx; a.ldy = ldy; a.ldq = ldq; a.M = M;
    a.ffn_chunks = d_hidden / CH;
    return dec_tail_launch(a, false, stream);
}

/* The same with the cross-attention's out_proj + residual + norm_cross in front (deformable_transformer.py:406-422): S = the sampled
 * rows, R = tgt in front of the block; image = gom_dec_tail_lin_image(out_proj) | the tail image with ALL THREE blocks in accumulator
 * order (gom_ffn_fused_image_acc_order for the FFN too). */
extern "C" int gom_dec_tail_proj_f32(const float* S, int lds, const float* R, int ldr, const void* image, int d_hidden,
                                     const float* p_inv_scale, const float* p_bias, const float* p_gamma, const float* p_beta,
                                     float p_eps, const float* w2_inv_scale, const float* b2, const float* gamma, const float* beta,
                                     float eps, const float* c_inv_scale, const float* c_b2, const float* W3, const float* b3,
                                     const float* ref, const float* dim_t128, const float* q_inv_scale, const float* q_b2, float* Y,
                                     int ldy, float* new_ref, float* qpos, int ldq, int M, int* flag, void* stream) {
    GOM_CHECK_ARG(S && R && image && p_inv_scale && p_bias && p_gamma && p_beta);
    GOM_CHECK_ARG(w2_inv_scale && b2 && gamma && beta && c_inv_scale && c_b2 && W3 && b3 && ref && Y && new_ref);
    GOM_CHECK_ARG(M >= 0 && d_hidden > 0 && (d_hidden % CH) == 0 && lds >= D && ldr >= D && ldy >= D && (lds % 4) == 0 &&
                  (ldr % 4) == 0 && (ldy % 4) == 0);
    GOM_CHECK_ARG(!qpos || (dim_t128 && q_inv_scale && q_b2 && ldq >= D && (ldq % 4) == 0 && GOM_ALIGNED16(qpos)));
    GOM_CHECK_ARG(GOM_ALIGNED16(S) && GOM_ALIGNED16(R) && GOM_ALIGNED16(Y) && GOM_ALIGNED16(image) && ((uintptr_t)new_ref % 8) == 0);
    GOM_CHECK_ARG(GOM_ALIGNED16(p_inv_scale) && GOM_ALIGNED16(p_bias) && GOM_ALIGNED16(p_gamma) && GOM_ALIGNED16(p_beta));
    GOM_CHECK_ARG(GOM_ALIGNED16(w2_inv_scale) && GOM_ALIGNED16(b2) && GOM_ALIGNED16(gamma) && GOM_ALIGNED16(beta) &&
                  GOM_ALIGNED16(c_inv_scale) && GOM_ALIGNED16(c_b2) && GOM_ALIGNED16(W3));
    GOM_CHECK_ARG(!qpos || (GOM_ALIGNED16(dim_t128) && GOM_ALIGNED16(q_inv_scale) && GOM_ALIGNED16(q_b2)));
    GOM_CHECK_ARG(Y != S && Y != R);                          // Y doubles as the kernel's parking space for the FFN's residual
    if (M == 0) return GOM_OK;
    TailArgs a{};
    a.X = S; a.R = R; a.ldr = ldr; a.p_s = p_inv_scale; a.p_b = p_bias; a.p_gamma = p_gamma; a.p_beta = p_beta; a.p_eps = p_eps;
    a.img = (const unsigned char*)image; a.s2 = w2_inv_scale; a.b2 = b2; a.gamma = gamma; a.beta = beta;
    a.c_s2 = c_inv_scale; a.c_b2 = c_b2; a.W3 = W3; a.b3 = b3; a.ref = ref; a.dim_t = dim_t128; a.q_s2 = q_inv_scale; a.q_b2 = q_b2;
    a.Y = Y; a.new_ref = new_ref; a.QP = qpos; a.flag = flag; a.eps = eps; a.ldx = lds; a.ldy = ldy; a.ldq = ldq; a.M = M;
    a.ffn_chunks = d_hidden / CH;
    return dec_tail_launch(a, true, stream);
}

