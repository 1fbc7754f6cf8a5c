// The row-local tail of a DeepSolo composite decoder layer, CU-COOPERATIVE form (round 6; same mathematics and the same chain as
// dec_tail.hip, /root/reference/third_party/adet/layers/deformable_transformer.py:406-422 out_proj + norm_cross, :352-354,368-369
// forward_ffn + norm3, :484-488 the reference refinement, :470-473 + adet/modeling/model/utils.py:24-37 the next layer's query
// position):
//
//   tgt3   = norm_cross(R + S Wo^T + bo)                       (optional)
//   tgt    = norm3(tgt3 + W2 relu(W1 tgt3 + b1) + b2)
//   ref'   = sigmoid(W3c relu(W2c relu(W1c tgt + b1c) + b2c) + b3c + inverse_sigmoid(ref))
//   qpos'  = W2r relu(W1r sine_embed(ref') + b1r) + b2r        (optional)
//
// dec_tail.hip gives every WAVE its own 32 rows and all 256 output columns.  At the decoder's M = frames x queries x points =
// 20 000 rows that is 1 250 sixteen-row MFMA groups on 1 024 SIMDs: the busiest SIMD holds two whatever the tile size, 157
// workgroups sit on 256 CUs, and each weight fragment read from LDS serves two row groups.  Here the four waves of a workgroup SHARE
// 80 rows (five row groups; 250 workgroups = one round of the chip) and split the OUTPUT COLUMNS:
//
//   * the rows live in LDS, not in registers: XP = the block's input as MFMA B-operand fragments (two fp16 planes, 8 k-steps x 5
//     row groups x 1 KB, fragment-linear: conflict-free ds_read_b128), HP = one chunk of 128 hidden units the same way;
//   * wave w computes hidden units 32 w .. 32 w + 31 of every chunk (GEMM1: its own two 16-row groups of W1 against ALL rows) and
//     output columns 64 w .. 64 w + 63 (GEMM2 / plain layers: four groups of W2 against all rows): 30 / 60 MFMAs per k-step and wave
//     for ten B fragments read from LDS (1 KB per 3 / 6 MFMAs; dec_tail.hip: 1 KB per 3) -- the busiest SIMD holds 1.25 row-group
//     equivalents instead of 2;
//   * the weights never touch LDS: each wave streams ITS quarter of a fragment-linear image straight from L2 into registers
//     (buffer_load_dwordx4, eight fragments per group, one group ahead), in consumption order, one linear stream per wave across
//     all blocks, so the stream never drains between them;
//   * a product comes out as Y^T (row = lane n of row group rg, lane group g holds columns 64 w + 16 cg + 4 g + e).  Residual, bias,
//     scale and LayerNorm run in that layout; the row statistics need the other three waves' columns: (sum, M2) per wave and row
//     through LDS, combined with Chan's formula (one barrier).  The finished values go back to XP as the next block's B operand by
//     ONE 16-byte store per lane, plane and (row group, k-step): k-slot j of lane group g at k-step s <-> feature
//     32 s + 16 (j >> 2) + 4 g + (j & 3) -- dec_tail.hip's accumulator order, baked into every weight image here (the prologue
//     writes the loaded rows in the same order);
//   * the 256 -> 2 layer: per-wave partial dot products through LDS, summed in wave order by every lane (deterministic); the sine
//     embedding is evaluated directly in operand order by the wave that owns the feature.
// Per row the arithmetic does not depend on what shares the launch (batch invariance); against dec_tail.hip the LayerNorm
// statistics (Chan's combination instead of two passes over the row) and nothing else differ: both are held to the same fp64
// tolerance by tests/test_dec_tail_gpu.py.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Range bookkeeping on what is SPLIT (gemm_f16x3.hip contract: an operand must stay within fp16's range, and a NaN must not pass):
// the main plane of a split value is fp16(x) -- infinite or NaN exactly when |x| >= 65520 or x is not finite -- so the running
// maximum of the planes' 15-bit magnitudes (v_and + v_pk_max_u16 per TWO values; non-negative values: the maximum alone) replaces a
// float maximum and a NaN detector per value.  Raised when a half reaches the exponent 0x1f.
__device__ __forceinline__ void t2_track(u16x2& m, const half8 plane0) {
    const u32x4 w = __builtin_bit_cast(u32x4, plane0);
#pragma unroll
    for (int i = 0; i < 4; ++i) m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, w[i] & 0x7fff7fffu));
}
__device__ __forceinline__ void t2_track_nonneg(u16x2& m, const half8 plane0) {
    const u32x4 w = __builtin_bit_cast(u32x4, plane0);
#pragma unroll
    for (int i = 0; i < 4; ++i) m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, w[i]));
}

__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int D = 256;                                   // model width
constexpr int RB = 80;                                   // rows per workgroup
constexpr int NRG = RB / 16;                             // MFMA row groups per workgroup
constexpr int HC = 128;                                  // hidden units per chunk (32 per wave)
constexpr int FRAG = 1024;
constexpr int XP_BYTES = 8 * NRG * 2 * FRAG;             // 80 KB: [k-step][row group][plane]
constexpr int HP_OFF = XP_BYTES;
constexpr int HP_BYTES = 4 * NRG * 2 * FRAG;             // 40 KB
constexpr int RED_OFF = HP_OFF + HP_BYTES;
constexpr int RED_BYTES = 2 * 8 * RB * 8;                // two buffers of [wave][row] float2 (up to eight waves)
constexpr int LDS_BYTES = RED_OFF + RED_BYTES;
// per wave and 256 -> 256 layer, and per wave and chunk of an MLP block (half W1, half W2): 8 weight groups of 32 / waves fragments
constexpr int wave_frags(int waves) { return 8 * (32 / waves); }

struct T2Args {
    const float* X;                                          // [M, 256]: tgt behind norm_cross, or (proj) the cross-attention's sampled rows
    const float* R;                                          // proj: tgt in front of the cross attention (norm_cross's residual)
    const unsigned char* img;
    const float *p_s, *p_b, *p_gamma, *p_beta;               // proj: 1 / row scale of Wo, bias, norm_cross
    const float *s1, *b1, *s2, *b2, *gamma, *beta;           // FFN: 1 / row scales and biases of W1 / W2, norm3
    const float *c_s1, *c_b1, *c_s2, *c_b2, *W3, *b3;        // ctrl_point_coord
    const float *ref, *dim_t;
    const float *q_s1, *q_b1, *q_s2, *q_b2;                  // ref_point_head
    float *Y, *new_ref, *QP;
    int* flag;
    float eps, p_eps;
    int ldx, ldr, ldy, ldq, M, ffn_chunks;
    unsigned wave_stride, img_bytes;
};

__device__ __forceinline__ float inv_sigmoid(float x) {   // adet/utils/misc.py:115-119, eps 1e-5 (as elementwise.hip)
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
    return logf(x1 / x2);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// sin and cos of an angle in [0, 2 pi] (dec_tail.hip: Cody-Waite quadrant reduction + the cephes single-precision kernels)
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

// schedule knobs (tools/dec_tail2_variants.py builds the file with other values): MFMAs between two of a step's eight weight loads
#ifndef T2_SPREAD1
#define T2_SPREAD1 3
#endif
#ifndef T2_SPREAD2
#define T2_SPREAD2 2
#endif

// -DT2_STAMPS (tools/dec_tail2_variants.py only): s_memtime at the phase boundaries of every wave into a buffer set by
// gom_dec_tail2_set_stamps -- [workgroup][wave][16] cycles since the wave's start; slots 10..12 = sums over the FFN's chunks
#ifdef T2_STAMPS
__device__ unsigned long long* g_t2_stamps = nullptr;
__device__ __forceinline__ unsigned long long t2_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define T2_STAMP(i)                                                                                        \
    if (g_t2_stamps && lane == 0) g_t2_stamps[((size_t)blockIdx.x * NW + wave) * 16 + (i)] = t2_clock() - t2_t0;
#define T2_STAMP_ADD(i, since)                                                                             \
    if (g_t2_stamps && lane == 0) g_t2_stamps[((size_t)blockIdx.x * NW + wave) * 16 + (i)] += t2_clock() - (since);
#define T2_NOW(var) const unsigned long long var = t2_clock();
#else
#define T2_STAMP(i)
#define T2_STAMP_ADD(i, since)
#define T2_NOW(var)
#endif

#define T2_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory")
#define T2_BARRIER() asm volatile("s_barrier" ::: "memory")

// NW = waves per workgroup: 4 (one per SIMD; a wave owns 64 output columns and 32 hidden units of a chunk) or 8 (TWO per SIMD; 32 and
// 16: half the accumulators, <= 256 registers -- one wave's epilogues and activation phases then run under the other's MFMAs)
template <bool WITH_QPOS, bool WITH_PROJ, int NW>
__global__ __launch_bounds__(64 * NW, 1) void dec_tail2_kernel(const T2Args p) {
    constexpr int CGN = 16 / NW;                             // output column groups per wave
    constexpr int HGN = 8 / NW;                              // hidden groups per wave and chunk
    constexpr int GF = 32 / NW;                              // fragments per weight group
    constexpr int WCOLS = 16 * CGN;                          // output columns per wave
    constexpr int UPW = RB / NW;                             // prologue units per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 15, fg = lane >> 4;
    const long tile0 = (long)blockIdx.x * RB;
#ifdef T2_STAMPS
    const unsigned long long t2_t0 = t2_clock();
#endif

    // ---- the wave's weight stream ----
    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, (int)p.img_bytes, 0x00020000);
    const int voff = lane * 16;
    int so = (int)(wave * p.wave_stride);                    // byte offset of the current block / chunk in the stream (uniform)
    half8 a0[GF], a1[GF], b0[10], b1[10];
#define T2_LOADA_(dst, grp)                                                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < GF; ++i_)                                                                       \
        dst[i_] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_img, voff, so + ((grp) * GF + i_) * FRAG, 0));
#ifdef T2_EXP_NO_A                                       /* timing experiment (wrong results): the weight stream is not read */
#define T2_LOADA(dst, grp)
#else
#define T2_LOADA(dst, grp) T2_LOADA_(dst, grp)
#endif
    const unsigned char* xp_lane = smem + lane * 16;
    const unsigned char* hp_lane = smem + HP_OFF + lane * 16;
#define T2_LOADB_(dst, base, s)                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) dst[i_] = *reinterpret_cast<const half8*>((base) + ((s) * 10 + i_) * FRAG);
#ifdef T2_EXP_NO_B                                       /* timing experiment (wrong results): the row fragments are read once */
#define T2_LOADB(dst, base, s)
#else
#define T2_LOADB(dst, base, s) T2_LOADB_(dst, base, s)
#endif
    T2_LOADA_(a0, 0)
#if defined(T2_EXP_NO_A) || defined(T2_EXP_NO_B)
    T2_LOADA_(a1, 1)
    T2_LOADB_(b0, xp_lane, 0)
    T2_LOADB_(b1, xp_lane, 1)
#endif

    // this lane's rows in the accumulator layout (row group rg: tile0 + 16 rg + fn), clamped for loads; tail rows are never stored
    long mrow[NRG];
    bool live[NRG];
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg) {
        const long m = tile0 + 16 * rg + fn;
        live[rg] = m < p.M;
        mrow[rg] = live[rg] ? m : p.M - 1;
    }

    float amax = 0.f, chk = 0.f;                             // range bookkeeping (gemm_f16x3.hip contract): the input rows, unsplit results
    u16x2 pmax = {0, 0};                                     // ... and every value split inside the kernel (t2_track)

    // ---- prologue: the 80 input rows -> XP.  80 units of (8 rows x 32 floats = one k-step) over the waves; a unit is ONE load
    //      instruction of eight whole 128-byte lines (lane: row l & 7, 16-byte piece l >> 3) and two 8-byte LDS stores per lane ----
    {
        f32x4 v[UPW];
        unsigned pf[NW == 8 ? 2 : 4];
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave * UPW + i, rb = u >> 3, part = u & 7;
            long m = tile0 + 8 * rb + (lane & 7);
            if (m > p.M - 1) m = p.M - 1;
            v[i] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * p.ldx + 32 * part + 4 * (lane >> 3));
        }
        // the whole image (3.4 MB) towards this XCD's L2: a weight group that misses stalls its wave for the trip to HBM.  BEHIND the
        // row loads (the vector-memory counter is in order: in front of them the rows wait for the image's lines, +7k cycles)
        __builtin_amdgcn_sched_barrier(0);
        gom_prefetch_image(p.img, p.img_bytes, tid, 64 * NW, pf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = wave * UPW + i, rb = u >> 3, part = u & 7;
            const int row = 8 * rb + (lane & 7), pp = lane >> 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
            unsigned h0, l0, h1, l1;
            gom_split2_f16(v[i][0], v[i][1], h0, l0);
            gom_split2_f16(v[i][2], v[i][3], h1, l1);
            unsigned char* dst = smem + ((part * NRG + (row >> 4)) * 2) * FRAG + ((row & 15) + 16 * (pp & 3)) * 16 + 8 * (pp >> 2);
            *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(dst + FRAG) = u32x2{l0, l1};
        }
        asm volatile("" : "+v"(amax));
        gom_prefetch_done(pf);
    }
    T2_BARRIER_LDS();
    T2_STAMP(0)

    f32x4 acc2[CGN][NRG];                                    // [column group of the wave][row group]
#define T2_ZERO_ACC2()                                                                                                      \
    _Pragma("unroll") for (int cg = 0; cg < CGN; ++cg)                                                                        \
        _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg) acc2[cg][rg] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 60 MFMAs: the wave's four column groups (fragments A[2 cg + plane]) against the five row groups (B[2 rg + plane]);
    // small products first (residual x main, main x residual, main x main), twenty accumulators between dependent ones
#define T2_MM2(A, B)                                                                                                        \
    _Pragma("unroll") for (int cg = 0; cg < CGN; ++cg)                                                                        \
        _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg) acc2[cg][rg] = mfma16(A[2 * cg + 1], B[2 * rg], acc2[cg][rg]);   \
    _Pragma("unroll") for (int cg = 0; cg < CGN; ++cg)                                                                        \
        _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg) acc2[cg][rg] = mfma16(A[2 * cg], B[2 * rg + 1], acc2[cg][rg]);   \
    _Pragma("unroll") for (int cg = 0; cg < CGN; ++cg)                                                                        \
        _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg) acc2[cg][rg] = mfma16(A[2 * cg], B[2 * rg], acc2[cg][rg]);
    // 30 MFMAs: the wave's two hidden groups (fragments A[4 half + 2 hg + plane]) against the five row groups
#define T2_MM1(A, half, B)                                                                                                  \
    _Pragma("unroll") for (int hg = 0; hg < HGN; ++hg)                                                                        \
        _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg)                                                                  \
            acc1[hg][rg] = mfma16(A[2 * HGN * (half) + 2 * hg + 1], B[2 * rg], acc1[hg][rg]);                                     \
    _Pragma("unroll") for (int hg = 0; hg < HGN; ++hg)                                                                        \
        _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg)                                                                  \
            acc1[hg][rg] = mfma16(A[2 * HGN * (half) + 2 * hg], B[2 * rg + 1], acc1[hg][rg]);                                     \
    _Pragma("unroll") for (int hg = 0; hg < HGN; ++hg)                                                                        \
        _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg)                                                                  \
            acc1[hg][rg] = mfma16(A[2 * HGN * (half) + 2 * hg], B[2 * rg], acc1[hg][rg]);
    // schedule pins: the step's ten LDS reads first, then its MFMAs with the eight weight loads spread between them
#define T2_PIN_B() __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#define T2_PIN_MA(n_mfma_per_load)                                                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < GF; ++q_) {                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, n_mfma_per_load, 0);                                                    \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                  \
    }
#define T2_PIN_M(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0);

    // ---- an MLP block: Y^T = W2 relu(W1 X^T / s1 + b1) over `nch` chunks of 128 hidden units.  On entry a0 = the block's first weight
    //      group (requested by the previous block), XP = the block's input (barrier passed). ----
#define T2_MLP(nch, inv1, bias1)                                                                                            \
    {                                                                                                                       \
        T2_ZERO_ACC2()                                                                                                      \
        T2_LOADB(b0, xp_lane, 0)                                                                                            \
        for (int c = 0; c < (nch); ++c) {                                                                                   \
            const bool more = c + 1 < (nch);                                                                                \
            T2_NOW(tc0_)                                                                                                    \
            f32x4 acc1[HGN][NRG];                                                                                           \
            _Pragma("unroll") for (int hg = 0; hg < HGN; ++hg)                                                                \
                _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg) acc1[hg][rg] = f32x4{0.f, 0.f, 0.f, 0.f};                \
            f32x4 sc[HGN], bi[HGN];                                                                                         \
            _Pragma("unroll") for (int hg = 0; hg < HGN; ++hg) {                                                              \
                const int h = HC * c + 16 * (HGN * wave + hg) + 4 * fg;                                                     \
                sc[hg] = *reinterpret_cast<const f32x4*>((inv1) + h);                                                       \
                bi[hg] = *reinterpret_cast<const f32x4*>((bias1) + h);                                                      \
            }                                                                                                               \
            T2_LOADB(b1, xp_lane, 1) T2_LOADA(a1, 1) T2_MM1(a0, 0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD1) T2_PIN_M(15 * HGN - GF * T2_SPREAD1)                  \
            T2_LOADB(b0, xp_lane, 2) T2_MM1(a0, 1, b1) T2_PIN_B() T2_PIN_M(15 * HGN)                                              \
            T2_LOADB(b1, xp_lane, 3) T2_LOADA(a0, 2) T2_MM1(a1, 0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD1) T2_PIN_M(15 * HGN - GF * T2_SPREAD1)                  \
            T2_LOADB(b0, xp_lane, 4) T2_MM1(a1, 1, b1) T2_PIN_B() T2_PIN_M(15 * HGN)                                              \
            T2_LOADB(b1, xp_lane, 5) T2_LOADA(a1, 3) T2_MM1(a0, 0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD1) T2_PIN_M(15 * HGN - GF * T2_SPREAD1)                  \
            T2_LOADB(b0, xp_lane, 6) T2_MM1(a0, 1, b1) T2_PIN_B() T2_PIN_M(15 * HGN)                                              \
            T2_LOADB(b1, xp_lane, 7) T2_LOADA(a0, 4) T2_MM1(a1, 0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD1) T2_PIN_M(15 * HGN - GF * T2_SPREAD1)                  \
            T2_MM1(a1, 1, b1) T2_PIN_M(15 * HGN)                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
            T2_STAMP_ADD(10, tc0_)                                                                                          \
            T2_NOW(tc1_)                                                                                                    \
            T2_BARRIER();                                            /* every wave is done with the previous chunk's HP */ \
            _Pragma("unroll") for (int rg = 0; rg < NRG; ++rg) {                                                            \
                f32x4 v[HGN];                                                                                               \
                _Pragma("unroll") for (int hg = 0; hg < HGN; ++hg)                                                          \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                           \
                        v[hg][e] = fmaxf(fmaf(acc1[hg][rg][e], sc[hg][e], bi[hg][e]), 0.f);                                 \
                if constexpr (HGN == 2) {                                                                                   \
                    half8 h0, h1;                                                                                           \
                    gom_split8_f16(v[0], v[1], h0, h1);                                                                     \
                    t2_track_nonneg(pmax, h0);                                                                              \
                    unsigned char* dst = smem + HP_OFF + ((wave * NRG + rg) * 2) * FRAG + lane * 16;                        \
                    *reinterpret_cast<half8*>(dst) = h0;                                                                    \
                    *reinterpret_cast<half8*>(dst + FRAG) = h1;                                                             \
                } else {              /* 16 hidden units per wave: its half of k-step wave >> 1's fragment, 8 bytes per lane */ \
                    unsigned h0, l0, h1, l1;                                                                                \
                    gom_split2_f16(v[0][0], v[0][1], h0, l0);                                                               \
                    gom_split2_f16(v[0][2], v[0][3], h1, l1);                                                               \
                    pmax = __builtin_elementwise_max(pmax, __builtin_elementwise_max(__builtin_bit_cast(u16x2, h0), __builtin_bit_cast(u16x2, h1))); \
                    unsigned char* dst = smem + HP_OFF + (((wave >> 1) * NRG + rg) * 2) * FRAG + lane * 16 + 8 * (wave & 1); \
                    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};                                                         \
                    *reinterpret_cast<u32x2*>(dst + FRAG) = u32x2{l0, l1};                                                  \
                }                                                                                                           \
            }                                                                                                               \
            T2_BARRIER_LDS();                                                                                               \
            T2_STAMP_ADD(11, tc1_)                                                                                          \
            T2_NOW(tc2_)                                                                                                    \
            T2_LOADB(b0, hp_lane, 0)                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
            T2_LOADB(b1, hp_lane, 1) T2_LOADA(a1, 5) T2_MM2(a0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)                     \
            T2_LOADB(b0, hp_lane, 2) T2_LOADA(a0, 6) T2_MM2(a1, b1) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)                     \
            T2_LOADB(b1, hp_lane, 3) T2_LOADA(a1, 7) T2_MM2(a0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)                     \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
            if (more) {                                                                                                     \
                T2_LOADB(b0, xp_lane, 0)                                                                                    \
            }                                                                                                               \
            T2_LOADA(a0, 8) T2_MM2(a1, b1) T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                              \
            T2_STAMP_ADD(12, tc2_)                                                                                          \
            so += 8 * GF * FRAG;                                                                                            \
        }                                                                                                                   \
    }

    // ---- row statistics of v = acc2 across the workgroup's 256 columns: per wave (sum, M2 about the wave's own mean) over its
    //      64 columns, exchanged through RED[buf], combined with Chan's formula.  One barrier. ----
    // (sums over the waves' contributions in a fixed pairwise order)
    auto tree = [&](const float (&v)[NW]) {
        if constexpr (NW == 4) return (v[0] + v[1]) + (v[2] + v[3]);
        else return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    };
    auto row_stats = [&](int buf, float eps, float (&mean)[NRG], float (&rstd)[NRG]) {
        float2* red = reinterpret_cast<float2*>(smem + RED_OFF) + buf * NW * RB;
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
            float s = 0.f;
#pragma unroll
            for (int cg = 0; cg < CGN; ++cg)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += acc2[cg][rg][e];
            const float mw = groups_sum(s) * (1.f / WCOLS);
            float q = 0.f;
#pragma unroll
            for (int cg = 0; cg < CGN; ++cg)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = acc2[cg][rg][e] - mw;
                    q = fmaf(d, d, q);
                }
            q = groups_sum(q);
            if (fg == 0) red[wave * RB + 16 * rg + fn] = make_float2(mw, q);
        }
        T2_BARRIER_LDS();
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
            float wm[NW], wq[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float2 w = red[k * RB + 16 * rg + fn];
                wm[k] = w.x;
                wq[k] = w.y;
            }
            const float m = tree(wm) * (1.f / NW);
            float m2 = tree(wq);
#pragma unroll
            for (int k = 0; k < NW; ++k) m2 = fmaf((float)WCOLS * (wm[k] - m), wm[k] - m, m2);
            mean[rg] = m;
            rstd[rg] = rsqrtf(m2 * (1.f / D) + eps);
        }
    };
    // the finished rows o[cg][rg] (accumulator layout) -> XP as the next block's B operand: k-steps 2 w, 2 w + 1
    auto to_xp = [&](const f32x4 (&o)[CGN][NRG]) {
#pragma unroll
        for (int sh = 0; sh < CGN / 2; ++sh)
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
                half8 h0, h1;
                gom_split8_f16(o[2 * sh][rg], o[2 * sh + 1][rg], h0, h1);
                t2_track(pmax, h0);
                unsigned char* dst = smem + ((((CGN / 2) * wave + sh) * NRG + rg) * 2) * FRAG + lane * 16;
                *reinterpret_cast<half8*>(dst) = h0;
                *reinterpret_cast<half8*>(dst + FRAG) = h1;
            }
    };

    // the FFN's residual in the accumulator layout: norm_cross's output (proj) or the input rows
    f32x4 res[CGN][NRG];

    // ================================ block 0: out_proj of the cross attention + norm_cross ================================
    if constexpr (WITH_PROJ) {
        T2_ZERO_ACC2()
        T2_LOADB(b0, xp_lane, 0)
        T2_LOADB(b1, xp_lane, 1) T2_LOADA(a1, 1) T2_MM2(a0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        T2_LOADB(b0, xp_lane, 2) T2_LOADA(a0, 2) T2_MM2(a1, b1) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        T2_LOADB(b1, xp_lane, 3) T2_LOADA(a1, 3) T2_MM2(a0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        T2_LOADB(b0, xp_lane, 4) T2_LOADA(a0, 4) T2_MM2(a1, b1) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        T2_LOADB(b1, xp_lane, 5) T2_LOADA(a1, 5) T2_MM2(a0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        T2_LOADB(b0, xp_lane, 6) T2_LOADA(a0, 6) T2_MM2(a1, b1) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        T2_LOADB(b1, xp_lane, 7) T2_LOADA(a1, 7) T2_MM2(a0, b0) T2_PIN_B() T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        T2_LOADA(a0, 8) T2_MM2(a1, b1) T2_PIN_MA(T2_SPREAD2) T2_PIN_M(15 * CGN - GF * T2_SPREAD2)
        __builtin_amdgcn_sched_barrier(0);
        so += 8 * GF * FRAG;
        T2_STAMP(1)
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) {
            const int col = WCOLS * wave + 16 * cg + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.p_s + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.p_b + col);
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
                // norm_cross's residual: tgt in front of the cross attention.  (Sixteen half-lines per instruction in this layout:
                // ~4.6k cycles of the CU's texture-address unit whenever it is issued -- requested under the loop's last step the
                // loop grew by what the epilogue saved, requested before the loop the in-order vector-memory counter held the
                // weight stream behind it: 9.4k -> 27k cycles.)
                const f32x4 xr = *reinterpret_cast<const f32x4*>(p.R + (size_t)mrow[rg] * p.ldr + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[cg][rg][e] = fmaf(acc2[cg][rg][e], sc[e], bi[e]) + xr[e];
            }
        }
        float mean[NRG], rstd[NRG];
        row_stats(0, p.p_eps, mean, rstd);
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) {
            const int col = WCOLS * wave + 16 * cg + 4 * fg;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.p_gamma + col);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.p_beta + col);
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                for (int e = 0; e < 4; ++e) res[cg][rg][e] = (acc2[cg][rg][e] - mean[rg]) * rstd[rg] * ga[e] + be[e];
        }
        to_xp(res);
        T2_BARRIER_LDS();
        T2_STAMP(2)
    }

    // ================================ block 1: the FFN + norm3 ================================
    T2_MLP(p.ffn_chunks, p.s1, p.b1)
    T2_STAMP(3)
    {
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) {
            const int col = WCOLS * wave + 16 * cg + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s2 + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.b2 + col);
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
                if constexpr (!WITH_PROJ) res[cg][rg] = *reinterpret_cast<const f32x4*>(p.X + (size_t)mrow[rg] * p.ldx + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[cg][rg][e] = fmaf(acc2[cg][rg][e], sc[e], bi[e]) + res[cg][rg][e];
            }
        }
        float mean[NRG], rstd[NRG];
        row_stats(1, p.eps, mean, rstd);
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) {
            const int col = WCOLS * wave + 16 * cg + 4 * fg;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + col);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + col);
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
#pragma unroll
                for (int e = 0; e < 4; ++e) res[cg][rg][e] = (acc2[cg][rg][e] - mean[rg]) * rstd[rg] * ga[e] + be[e];
                if (live[rg]) *reinterpret_cast<f32x4*>(p.Y + (size_t)mrow[rg] * p.ldy + col) = res[cg][rg];
            }
        }
        to_xp(res);
        T2_BARRIER_LDS();
        T2_STAMP(4)
    }

    // ================================ block 2: ctrl_point_coord + the reference refinement ================================
    T2_MLP(2, p.c_s1, p.c_b1)
    T2_STAMP(5)
    float nref[NRG][2];
    {
        float2* red = reinterpret_cast<float2*>(smem + RED_OFF);     // buffer 0 (its last readers passed the FFN's barriers)
        float dx[NRG], dy[NRG];
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) dx[rg] = dy[rg] = 0.f;
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) {
            const int col = WCOLS * wave + 16 * cg + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.c_s2 + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.c_b2 + col);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.W3 + col);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(p.W3 + D + col);
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = fmaf(acc2[cg][rg][e], sc[e], bi[e]);
                    chk = fmaf(h, 0.f, chk);                 // (in front of the ReLU: max(NaN, 0) = 0 would hide it)
                    const float hr = fmaxf(h, 0.f);
                    dx[rg] = fmaf(hr, w0[e], dx[rg]);
                    dy[rg] = fmaf(hr, w1[e], dy[rg]);
                }
        }
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
            const float sx = groups_sum(dx[rg]), sy = groups_sum(dy[rg]);
            if (fg == 0) red[wave * RB + 16 * rg + fn] = make_float2(sx, sy);
        }
        T2_BARRIER_LDS();
        const float bx = p.b3[0], by = p.b3[1];
#pragma unroll
        for (int rg = 0; rg < NRG; ++rg) {
            float wx[NW], wy[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float2 w = red[k * RB + 16 * rg + fn];
                wx[k] = w.x;
                wy[k] = w.y;
            }
            const float ddx = tree(wx) + bx, ddy = tree(wy) + by;
            const float rx0 = p.ref[mrow[rg] * 2], ry0 = p.ref[mrow[rg] * 2 + 1];
            nref[rg][0] = sigmoidf(ddx + inv_sigmoid(rx0));
            nref[rg][1] = sigmoidf(ddy + inv_sigmoid(ry0));
            if (live[rg] && fg == 0 && wave == 0) *reinterpret_cast<f32x2*>(p.new_ref + mrow[rg] * 2) = f32x2{nref[rg][0], nref[rg][1]};
        }
        asm volatile("" : "+v"(chk));
    }

    if constexpr (WITH_QPOS) {
        // ---- the next layer's point embedding (gen_point_pos_embed: channels [0, 128) <- x, [128, 256) <- y, sin on even, cos on odd
        //      channels of a pair that shares dim_t), evaluated by the wave that owns the feature, straight into XP ----
        {
            f32x4 o[CGN][NRG];
#pragma unroll
            for (int cg = 0; cg < CGN; ++cg) {
                const f32x4 dt = *reinterpret_cast<const f32x4*>(p.dim_t + ((WCOLS * wave + 16 * cg + 4 * fg) & 127));
                // pos = pts * 2 pi / dim_t (utils.py:24-37) with the quotient as a product by 1 / dim_t refined to fp32 accuracy (one
                // Newton step on v_rcp_f32): two reciprocals per column group instead of ten divisions (~10 instructions each)
                float rdt[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float r0 = __builtin_amdgcn_rcpf(dt[2 * k]);
                    rdt[k] = fmaf(r0, fmaf(-dt[2 * k], r0, 1.f), r0);
                }
#pragma unroll
                for (int rg = 0; rg < NRG; ++rg) {
                    const float e = nref[rg][(WCOLS * wave) >> 7] * 6.283185307179586f;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {            // channels (4 g + 2 k, 4 g + 2 k + 1) of the quad: one angle
                        float sn, cs;
                        sincos_0_2pi(e * rdt[k], sn, cs);
                        o[cg][rg][2 * k] = sn;
                        o[cg][rg][2 * k + 1] = cs;
                    }
                }
            }
            to_xp(o);
            T2_BARRIER_LDS();
            T2_STAMP(6)
        }
        // ================================ block 3: ref_point_head ================================
        T2_MLP(2, p.q_s1, p.q_b1)
        T2_STAMP(7)
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) {
            const int col = WCOLS * wave + 16 * cg + 4 * fg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.q_s2 + col);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(p.q_b2 + col);
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = fmaf(acc2[cg][rg][e], sc[e], bi[e]);
                    chk = fmaf(o[e], 0.f, chk);
                }
                if (live[rg]) *reinterpret_cast<f32x4*>(p.QP + (size_t)mrow[rg] * p.ldq + col) = o;
            }
        }
        asm volatile("" : "+v"(chk));
    }
    T2_STAMP(8)
    // an operand left fp16's range, or a result is not finite (gemm_f16x3.hip contract; fmaxf drops a NaN, `chk` catches it)
    if ((!(amax <= 65504.f) || pmax[0] >= 0x7c00 || pmax[1] >= 0x7c00 || !(chk == 0.f)) && p.flag) atomicOr(p.flag, 1);
#undef T2_LOADA
#undef T2_LOADB
#undef T2_ZERO_ACC2
#undef T2_MM1
#undef T2_MM2
#undef T2_PIN_B
#undef T2_PIN_MA
#undef T2_PIN_M
#undef T2_MLP
}

// ---- weight images: per wave w (of nw = 4 or 8) one linear stream; element j of lane (m, kg) of a fragment holds plane p of a
//      row-scaled weight (gom_split_f16x2) at row 16 t + m and input index perm(s, kg, j) = 32 s + 16 (j >> 2) + 4 kg + (j & 3).
//      cgn = 16 / nw output column groups and hgn = 8 / nw hidden groups per wave, gf = 32 / nw fragments per weight group ----
// a 256 -> 256 layer: fragment q = gf s + 2 cg + p (s = 0..7 k-step, cg < cgn): plane p of W[16 (cgn w + cg) + m][perm(s, kg, j)]
__global__ __launch_bounds__(256) void t2_lin_image_kernel(const unsigned short* __restrict__ planes, long ps, int ld, int nw,
                                                           unsigned short* __restrict__ img, long wave_stride_el) {
    const int lf = wave_frags(nw), gf = 32 / nw, cgn = 16 / nw;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)nw * lf * 512) return;
    const int e = (int)(i % 512), q = (int)((i / 512) % lf), w = (int)(i / (512L * lf));
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    const int s = q / gf, cg = (q % gf) >> 1, pl = q & 1;
    img[w * wave_stride_el + (long)q * 512 + e] =
        planes[pl * ps + (size_t)(16 * (cgn * w + cg) + m) * ld + 32 * s + 16 * (j >> 2) + 4 * kg + (j & 3)];
}
// an MLP block of F hidden units: per chunk c of 128, fragments q = cf c + (2 hgn s + 2 hg + p | cf / 2 + gf s2 + 2 cg + p), cf = 8 gf:
//   W1 part: plane p of W1[128 c + 16 (hgn w + hg) + m][perm(s, kg, j)]
//   W2 part: plane p of W2[16 (cgn w + cg) + m][128 c + perm(s2, kg, j)]
__global__ __launch_bounds__(256) void t2_mlp_image_kernel(const unsigned short* __restrict__ p1, long ps1, int ld1,
                                                           const unsigned short* __restrict__ p2, long ps2, int ld2, int F, int nw,
                                                           unsigned short* __restrict__ img, long wave_stride_el) {
    const int cf = wave_frags(nw), gf = 32 / nw, cgn = 16 / nw, hgn = 8 / nw;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_wave = (long)(F / HC) * cf * 512;
    if (i >= nw * per_wave) return;
    const int e = (int)(i % 512), w = (int)(i / per_wave);
    const long qq = (i % per_wave) / 512;
    const int c = (int)(qq / cf), q = (int)(qq % cf);
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    unsigned short v;
    if (q < cf / 2) {
        const int s = q / (2 * hgn), hg = (q % (2 * hgn)) >> 1, pl = q & 1;
        v = p1[pl * ps1 + (size_t)(HC * c + 16 * (hgn * w + hg) + m) * ld1 + 32 * s + 16 * (j >> 2) + 4 * kg + (j & 3)];
    } else {
        const int r = q - cf / 2, s2 = r / gf, cg = (r % gf) >> 1, pl = r & 1;
        v = p2[pl * ps2 + (size_t)(16 * (cgn * w + cg) + m) * ld2 + HC * c + 32 * s2 + 16 * (j >> 2) + 4 * kg + (j & 3)];
    }
    img[w * wave_stride_el + qq * 512 + e] = v;
}

}  // namespace

/* bytes of ONE wave's stream (= the stride between the waves' streams); the image is `waves` x this.  waves = 4 or 8. */
extern "C" long gom_dec_tail2_wave_bytes(int d_model, int d_hidden, int with_proj, int with_qpos, int waves) {
    if (d_model != D || d_hidden <= 0 || (d_hidden % HC) != 0 || (waves != 4 && waves != 8)) return -1;
    return (long)((with_proj ? 1 : 0) + d_hidden / HC + 2 + (with_qpos ? 2 : 0)) * wave_frags(waves) * FRAG;
}

extern "C" int gom_dec_tail2_image_lin(const void* w_planes, long w_plane_stride, int ld, void* image, long wave_bytes, long offset_bytes,
                                       int waves, void* stream) {
    GOM_CHECK_ARG(waves == 4 || waves == 8);
    const int lf = wave_frags(waves);
    GOM_CHECK_ARG(w_planes && image && ld >= D && wave_bytes >= offset_bytes + (long)lf * FRAG && (offset_bytes % FRAG) == 0);
    hipLaunchKernelGGL(t2_lin_image_kernel, dim3((unsigned)cdiv((long)waves * lf * 512, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w_planes, w_plane_stride, ld, waves, (unsigned short*)((unsigned char*)image + offset_bytes),
                       wave_bytes / 2);
    return gom_launch_status();
}

extern "C" int gom_dec_tail2_image_mlp(const void* w1_planes, long w1_plane_stride, int ld1, const void* w2_planes, long w2_plane_stride,
                                       int ld2, int d_hidden, void* image, long wave_bytes, long offset_bytes, int waves, void* stream) {
    GOM_CHECK_ARG(waves == 4 || waves == 8);
    const int cf = wave_frags(waves);
    GOM_CHECK_ARG(w1_planes && w2_planes && image && d_hidden > 0 && (d_hidden % HC) == 0 && ld1 >= D && ld2 >= d_hidden);
    GOM_CHECK_ARG(wave_bytes >= offset_bytes + (long)(d_hidden / HC) * cf * FRAG && (offset_bytes % FRAG) == 0);
    const long total = (long)waves * (d_hidden / HC) * cf * 512;
    hipLaunchKernelGGL(t2_mlp_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w1_planes, w1_plane_stride, ld1, (const unsigned short*)w2_planes, w2_plane_stride, ld2,
                       d_hidden, waves, (unsigned short*)((unsigned char*)image + offset_bytes), wave_bytes / 2);
    return gom_launch_status();
}

#ifdef T2_STAMPS
extern "C" int gom_dec_tail2_set_stamps(void* device_buffer) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_t2_stamps), &device_buffer, sizeof(void*));
}
#endif

#define GOM_ALIGNED16(ptr) (((uintptr_t)(ptr) % 16) == 0)

/* S / R / p_*: NULL = no out_proj block in front (X = S is then tgt behind norm_cross).  qpos NULL = last layer. */
extern "C" int gom_dec_tail2_f32(const float* S, int lds, const float* R, int ldr, const void* image, long wave_bytes, int d_hidden,
                                 const float* p_inv_scale, const float* p_bias, const float* p_gamma, const float* p_beta, float p_eps,
                                 const float* w1_inv_scale, const float* b1, const float* w2_inv_scale, const float* b2,
                                 const float* gamma, const float* beta, float eps, const float* c_inv1, const float* c_b1,
                                 const float* c_inv2, const float* c_b2, const float* W3, const float* b3, const float* ref,
                                 const float* dim_t128, const float* q_inv1, const float* q_b1, const float* q_inv2, const float* q_b2,
                                 float* Y, int ldy, float* new_ref, float* qpos, int ldq, int M, int waves, int* flag, void* stream) {
    const bool proj = R != nullptr;
    GOM_CHECK_ARG(waves == 4 || waves == 8);
    GOM_CHECK_ARG(S && image && w1_inv_scale && b1 && w2_inv_scale && b2 && gamma && beta && c_inv1 && c_b1 && c_inv2 && c_b2 && W3 && b3 &&
                  ref && Y && new_ref);
    GOM_CHECK_ARG(!proj || (p_inv_scale && p_bias && p_gamma && p_beta && ldr >= D && (ldr % 4) == 0 && GOM_ALIGNED16(R) &&
                            GOM_ALIGNED16(p_inv_scale) && GOM_ALIGNED16(p_bias) && GOM_ALIGNED16(p_gamma) && GOM_ALIGNED16(p_beta)));
    GOM_CHECK_ARG(M >= 0 && d_hidden > 0 && (d_hidden % HC) == 0 && lds >= D && ldy >= D && (lds % 4) == 0 && (ldy % 4) == 0);
    GOM_CHECK_ARG(!qpos || (dim_t128 && q_inv1 && q_b1 && q_inv2 && q_b2 && ldq >= D && (ldq % 4) == 0 && GOM_ALIGNED16(qpos) &&
                            GOM_ALIGNED16(dim_t128) && GOM_ALIGNED16(q_inv1) && GOM_ALIGNED16(q_b1) && GOM_ALIGNED16(q_inv2) &&
                            GOM_ALIGNED16(q_b2)));
    GOM_CHECK_ARG(GOM_ALIGNED16(S) && GOM_ALIGNED16(Y) && GOM_ALIGNED16(image) && ((uintptr_t)new_ref % 8) == 0);
    GOM_CHECK_ARG(GOM_ALIGNED16(w1_inv_scale) && GOM_ALIGNED16(b1) && GOM_ALIGNED16(w2_inv_scale) && GOM_ALIGNED16(b2) &&
                  GOM_ALIGNED16(gamma) && GOM_ALIGNED16(beta) && GOM_ALIGNED16(c_inv1) && GOM_ALIGNED16(c_b1) && GOM_ALIGNED16(c_inv2) &&
                  GOM_ALIGNED16(c_b2) && GOM_ALIGNED16(W3));
    GOM_CHECK_ARG(wave_bytes == gom_dec_tail2_wave_bytes(D, d_hidden, proj ? 1 : 0, qpos ? 1 : 0, waves) && waves * wave_bytes < (1L << 31));
    if (M == 0) return GOM_OK;
    T2Args a{};
    a.X = S; a.R = R; a.img = (const unsigned char*)image;
    a.p_s = p_inv_scale; a.p_b = p_bias; a.p_gamma = p_gamma; a.p_beta = p_beta; a.p_eps = p_eps;
    a.s1 = w1_inv_scale; a.b1 = b1; a.s2 = w2_inv_scale; a.b2 = b2; a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.c_s1 = c_inv1; a.c_b1 = c_b1; a.c_s2 = c_inv2; a.c_b2 = c_b2; a.W3 = W3; a.b3 = b3; a.ref = ref; a.dim_t = dim_t128;
    a.q_s1 = q_inv1; a.q_b1 = q_b1; a.q_s2 = q_inv2; a.q_b2 = q_b2;
    a.Y = Y; a.new_ref = new_ref; a.QP = qpos; a.flag = flag;
    a.ldx = lds; a.ldr = ldr; a.ldy = ldy; a.ldq = ldq; a.M = M; a.ffn_chunks = d_hidden / HC;
    a.wave_stride = (unsigned)wave_bytes; a.img_bytes = (unsigned)(waves * wave_bytes);
    const void* k[8] = {(const void*)dec_tail2_kernel<false, false, 4>, (const void*)dec_tail2_kernel<true, false, 4>,
                        (const void*)dec_tail2_kernel<false, true, 4>, (const void*)dec_tail2_kernel<true, true, 4>,
                        (const void*)dec_tail2_kernel<false, false, 8>, (const void*)dec_tail2_kernel<true, false, 8>,
                        (const void*)dec_tail2_kernel<false, true, 8>, (const void*)dec_tail2_kernel<true, true, 8>};
    for (int i = 0; i < 8; ++i) {
        hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    }
    const dim3 grid((unsigned)cdiv(M, RB)), block(64 * waves);
    hipStream_t s = (hipStream_t)stream;
#define T2_LAUNCH(NW)                                                                                            \
    if (proj) {                                                                                                  \
        if (qpos) hipLaunchKernelGGL((dec_tail2_kernel<true, true, NW>), grid, block, LDS_BYTES, s, a);          \
        else hipLaunchKernelGGL((dec_tail2_kernel<false, true, NW>), grid, block, LDS_BYTES, s, a);              \
    } else {                                                                                                     \
        if (qpos) hipLaunchKernelGGL((dec_tail2_kernel<true, false, NW>), grid, block, LDS_BYTES, s, a);         \
        else hipLaunchKernelGGL((dec_tail2_kernel<false, false, NW>), grid, block, LDS_BYTES, s, a);             \
    }
    if (waves == 8) { T2_LAUNCH(8) } else { T2_LAUNCH(4) }
#undef T2_LAUNCH
    return gom_launch_status();
}
