// Fused transformer FFN block on the fp16 matrix cores (f16x3 split, fp32-class accuracy):
//
//   Y = LayerNorm( X + relu(X W1^T + b1) W2^T + b2 ) * gamma + beta        X, Y [M, 256] fp32, hidden width F
//
// = `linear1 -> ReLU -> linear2 -> +residual -> norm` of every DeepSolo encoder / decoder layer
// (/root/reference/third_party/adet/layers/deformable_transformer.py:250-251,266-273 encoder `forward_ffn` + norm2,
//  :352-354,368-369 decoder `forward_ffn` + norm3).  As three launches (GEMM, GEMM, LayerNorm) the block moves
// 1 + 4 | 4 + 1 + 1 | 1 + 1 = 13 KB per row through HBM for 1.05 MFLOP per row -- 80 FLOP/B, below the chip's balance, i.e.
// HBM-bound at ~500 TFLOP/s even with perfect kernels.  Fused, the hidden activations never leave the CU: 2 KB per row.
//
// Structure (one workgroup = 4 waves = 128 rows, one wave per SIMD with the whole 512-register file):
//   * each wave owns 32 rows = two row groups of 16.  Their two fp16 planes (x = x0 + x1, 22 significand bits: gemm_f16x3.hip)
//     stay in 128 VGPRs for the whole kernel, already in the MFMA operand layout;
//   * MFMA shape v_mfma_f32_16x16x32_f16: on random data the chip holds a higher clock under it than under 32x32x16 at equal
//     cycles per FLOP -- 1.22x the FLOP/s in a bare loop with this kernel's operand traffic, 1.3x with its VALU share beside
//     it (tools/exp/mfma_shape_probe.py; MI355X_MICROARCH.md "DVFS give-back" item 7).  Same fragments per chunk, same LDS
//     bytes per FLOP, same register budget; a k-step is 32 wide, so per output element the plane products are summed per 32
//     (not 16) inputs: fp32-class like the tile kernel's order, not its bits (tests hold both to the same fp64 tolerance);
//   * both products are computed TRANSPOSED so that the row of X is the LANE of every accumulator:
//       H^T[32 hidden x 32 rows]  = W1c[32 x 256] . X^T        (A = weight fragment from LDS, B = X fragment in registers)
//       Y^T[256 x 32 rows]       += W2[:, chunk] . H^T          (A = weight fragment from LDS, B = H^T)
//     The accumulators of the first product ARE the B operand of the second once converted to fp16: lane (row n, group g) holds
//     hidden units 16 Hh + 4 g + i of the chunk (Hh = 0, 1; i = 0..3) -- exactly the eight k-slots the 32-wide k-step of the
//     second product wants from it (cdna_hip_programming.md §3, "An accumulator tile as the next MFMA's operand"): no LDS round
//     trip, no shuffle.  The k order that implies (slot j of lane group g <-> hidden unit 16 (j >> 2) + 4 g + (j & 3)) is baked
//     into the weight image, which costs nothing: the weights are constants;
//   * the weights (2 x 2 planes x F x 256 fp16 = 2 MB at F = 1024, L2-resident) stream through a two-stage LDS ring by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`), one 65 KB stage per 32 hidden units.  The image is FRAGMENT-LINEAR: every MFMA operand
//     fragment is one contiguous KB in the order the lanes read it, so the DMA is a linear copy and every ds_read_b128 is
//     conflict-free by construction;
//   * epilogue: Y^T goes through the (now free) LDS ring to row-major, then 16 lanes per row apply the weight scale, bias,
//     residual and the LayerNorm of norm.hip (same two-pass arithmetic) and store whole 1 KB rows.
// MFMA work per 128 rows: 4 waves x 6144 v_mfma_f32_16x16x32_f16; LDS reads 1/3 KB per MFMA; L2 -> LDS 2 MB.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int D = 256;                                   // model width (fixed: every shipped config)
constexpr int CH = 32;                                   // hidden units per weight chunk
constexpr int FRAG = 1024;                               // bytes of one MFMA operand fragment (64 lanes x 8 fp16)
constexpr int W1_FRAGS = (D / 32) * (CH / 16) * 2;       // k-steps of 32 x hidden groups of 16 x planes
constexpr int W2_FRAGS = (D / 16) * 2;                   // output groups of 16 x planes (ONE k-step: the chunk's 32 hidden units)
constexpr int STAGE_FRAGS = W1_FRAGS + W2_FRAGS + 1;     // + one fragment of (1 / row scale, bias) of the chunk
constexpr int STAGE_BYTES = STAGE_FRAGS * FRAG;
constexpr int BM = 128;
constexpr int LDS_BYTES = 2 * STAGE_BYTES > BM * D * 4 ? 2 * STAGE_BYTES : BM * D * 4;

struct FfnArgs {
    const float* X;
    const unsigned char* img;
    const float* s2;
    const float* b2;
    const float* gamma;
    const float* beta;
    float* Y;
    int* flag;
    float eps;
    int ldx, ldy, M, chunks, stagger;
    long row_base;                                           // first row of this launch's tiles (the half-height tail launch)
    int relu_out;                                            // PLAIN form: ReLU behind the second layer
};

// sum over the 16 lanes of a DPP row, result in every lane: quad swaps (xor 1, xor 2), then the two mirrors
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, int ctrl_tag) {
        const int xi = __builtin_bit_cast(int, x);
        int r;
        if (ctrl_tag == 0) r = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
        else if (ctrl_tag == 1) r = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        else if (ctrl_tag == 2) r = __builtin_amdgcn_update_dpp(0, xi, 0x141, 0xF, 0xF, true);  // row_half_mirror
        else r = __builtin_amdgcn_update_dpp(0, xi, 0x140, 0xF, 0xF, true);                     // row_mirror
        return __builtin_bit_cast(float, r);
    };
    v += dpp(v, 0);
    v += dpp(v, 1);
    v += dpp(v, 2);
    v += dpp(v, 3);
    return v;
}

__device__ __forceinline__ void split2(float x, float y, unsigned int& q0, unsigned int& q1) { gom_split2_f16(x, y, q0, q1); }

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, half8& p0, half8& p1) {
    unsigned int l0, l1, l2, l3, h0, h1, h2, h3;
    split2(a[0], a[1], l0, h0);
    split2(a[2], a[3], l1, h1);
    split2(b[0], b[1], l2, h2);
    split2(b[2], b[3], l3, h3);
    p0 = __builtin_bit_cast(half8, (u32x4{l0, l1, l2, l3}));
    p1 = __builtin_bit_cast(half8, (u32x4{h0, h1, h2, h3}));
}

// One KB of the weight stream straight into LDS (buffer_load_dwordx4 ... lds).  The MUBUF form, not global_load_lds: hipcc
// books a FLAT-segment LDS-DMA as "may return out of order" and from then on turns every counted s_waitcnt lgkmcnt(N) of the
// loop into lgkmcnt(0), which serialises the fragment prefetch below (measured: no gain over three launches).
__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned byte_offset, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)byte_offset, 0, 0, 0);
}

// PLAIN = true: a two-layer perceptron  Y = [relu](relu(X W1^T + b1) W2^T + b2)  -- no residual, no LayerNorm -- on the same
// pipeline: the decoder's ref_point_head (deformable_transformer.py:470-473, adet/modeling/model/utils.py MLP) and the first two
// layers of its three-layer coordinate / boundary heads (:484-488, detection_transformer_wobackbone.py:238-253) at
// M = frames x queries x points rows, where two launches of the row-resident GEMM cost two workgroup latencies.
// RG = row groups of 16 per wave: 2 = the 128-row tile; 1 = a HALF-HEIGHT tile of 64 rows for the last, partly filled round of
// a long launch (2 324 tiles on 256 CUs = 9.08 rounds: the 20 tiles of the tenth round cost a whole round).  A tile's time is
// its MFMAs or its weight stream's issue, whichever is longer; half the rows halve the first, so 39 half-height tiles end the
// launch ~half a round earlier.  Rows are independent and every row's arithmetic is the same: the same bits.
template <bool PLAIN, int RG = 2>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(const FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 15, fg = lane >> 4;                // row inside a row group | k-group of an operand = feature quad of a result
    constexpr int WR = 16 * RG, BMT = 4 * WR;                // rows per wave, rows per tile
    const long tile0 = p.row_base + (long)blockIdx.x * BMT;
    const long row0 = tile0 + wave * WR;

    // Long launches (>= 4 rounds of workgroups): the first round starts STAGGERED, by up to 7 x stagger sleeps of ~0.4 us over
    // the CUs of an XCD.  With one workgroup per CU and equal durations every CU otherwise reaches its prologue, its weight
    // stream and its epilogue at the same moment, round after round, and HBM / L2 see bursts with idle time between them;
    // desynchronised, the memory phases of one CU run under the MFMA phases of the others.  Measured in the step (same box,
    // alternating runs): this kernel 575 -> 572 us per launch on average, proj_ln.hip 96 -> 92 us; in a loop of its own launches
    // the kernel gains 10 % (1046 -> 944 us at M = 297k), which the step does not see.
    // a one-round launch (the decoder's two-layer perceptrons): the image, cold between the layers of a step, towards L2 first
    unsigned pf[2] = {0u, 0u};
    if (gridDim.x <= 256) gom_prefetch_image(p.img, (unsigned)(p.chunks * STAGE_BYTES), tid, 256, pf);
    if (p.stagger > 0 && blockIdx.x < 256)
        for (int i = 0; i < (int)((blockIdx.x >> 3) & 7) * p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    const __amdgpu_buffer_rsrc_t rs_img =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, p.chunks * STAGE_BYTES, 0x00020000);
    auto dma_stage = [&](int c, int stage) {                 // 65 fragments, dealt to the four waves
        const unsigned src = (unsigned)c * STAGE_BYTES + lane * 16;
        unsigned char* dst = smem + stage * STAGE_BYTES;
        for (int f = wave; f < STAGE_FRAGS; f += 4) dma_fragment(rs_img, src + f * FRAG, dst + f * FRAG);
    };

    // ---- this wave's 32 rows of X as B-operand fragments: lane (n, kg) holds X[row 16 R + n][32 s + 8 kg .. + 7] in xf[.][8 R + s],
    // two planes (whole-line loads + a layout change in the ring's second slot, which the first stage does not use: common.h)
    int range_bad = 0;                                       // an operand beyond fp16's range (gemm_f16x3.hip contract)
    half8 xf[2][D / 16];
    {
        float xmax = 0.f;
        auto xrow = [&](int r) {
            long m = row0 + (RG == 2 ? r : (r & 15));          // (half-height: the unused second row group repeats the first)
            if (m > p.M - 1) m = p.M - 1;                     // tail rows recompute the last row (never stored)
            return p.X + (size_t)m * p.ldx;
        };
        gom_rows_to_fragments_t<128, false, true>(xrow, xrow, reinterpret_cast<float*>(smem + STAGE_BYTES) + wave * (32 * 128), lane, xf,
                                                  xmax, [&]() { dma_stage(0, 0); });
        range_bad = !(xmax <= 65504.f);
    }

    f32x4 acc2[D / 16][RG];                                  // [output group of 16][row group]
#pragma unroll
    for (int t = 0; t < D / 16; ++t)
#pragma unroll
        for (int r = 0; r < RG; ++r) acc2[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gom_prefetch_done(pf);
    __syncthreads();

    float hmax = 0.f;                                        // largest hidden activation of this lane's rows
    constexpr unsigned OOB = 0x7FFF0000u;                    // beyond num_records: such a DMA writes zeros (into an unused stage)
    for (int c = 0; c < p.chunks; ++c) {
        const int st = c & 1;
        // The next stage's 65 fragments: fragment 64 (scale | bias) by wave 0 here, fragments wave, wave + 4, ..., wave + 60 --
        // sixteen per wave -- ONE PER SIX MFMAs inside the products below.  An LDS-DMA instruction costs its wave 100-140
        // cycles of issue (s_memtime stamps, gemm_k256.hip); issued in a block in front of the MFMAs that was ~2200 cycles per
        // chunk with the matrix pipe idle (one wave per SIMD), beside running MFMAs it is hidden.
        const bool more = c + 1 < p.chunks;
        if (more && wave == 0)
            dma_fragment(rs_img, (unsigned)(c + 1) * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG + lane * 16,
                         smem + (st ^ 1) * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG);
        const unsigned nsrc = more ? (unsigned)(c + 1) * STAGE_BYTES + wave * FRAG + lane * 16 : OOB;
        unsigned char* ndst = smem + (st ^ 1) * STAGE_BYTES + wave * FRAG;
#define FFN_DMA(i) dma_fragment(rs_img, nsrc + (i) * 4 * FRAG, ndst + (i) * 4 * FRAG);
        const unsigned char* base = smem + st * STAGE_BYTES + lane * 16;

        // The 64 weight fragments of the chunk are consumed in eight groups of eight; with ONE wave per SIMD nothing else hides
        // the LDS latency, so group g + 1 is read into the other register set BEFORE the 24 MFMAs of group g are issued
        // (explicit two-deep software pipeline; the sched_group_barrier pairs pin that order -- left alone the compiler issues
        // every fragment read right in front of its MFMAs: reads, wait, MFMAs, ... = 28 % MFMA busy).
        half8 fa[8], fb[8];
#define FFN_LOAD(dst, g)                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
        dst[i_] = *reinterpret_cast<const half8*>(base + ((g) * 8 + i_) * FRAG);
#define FFN_PIN3()                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#define FFN_PIN1()                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);   \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#define FFN_PIN0()                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        // ---- H^T chunk = W1c . X^T : 2 hidden groups x 2 row groups of 16 x 16, 8 k-steps x 3 plane products (smallest first);
        //      fragment 4 i + 2 Hh + p of a group = plane p of hidden group Hh at the group's k-step i ----
        f32x4 acc1[2][RG];
#pragma unroll
        for (int h_ = 0; h_ < 2; ++h_)
#pragma unroll
            for (int r_ = 0; r_ < RG; ++r_) acc1[h_][r_] = f32x4{0.f, 0.f, 0.f, 0.f};
#define FFN_GEMM1(src, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                        \
        const int s_ = (g) * 2 + i_;                                                                          \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < RG; ++r_)                                                 \
                acc1[h_][r_] = mfma16(src[4 * i_ + 2 * h_ + 1], xf[0][8 * r_ + s_], acc1[h_][r_]);            \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < RG; ++r_)                                                 \
                acc1[h_][r_] = mfma16(src[4 * i_ + 2 * h_], xf[1][8 * r_ + s_], acc1[h_][r_]);                \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                      \
            _Pragma("unroll") for (int r_ = 0; r_ < RG; ++r_)                                                 \
                acc1[h_][r_] = mfma16(src[4 * i_ + 2 * h_], xf[0][8 * r_ + s_], acc1[h_][r_]);                \
    }
        FFN_LOAD(fa, 0)
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // group 0's reads come first, then (reads, MFMAs) pairs
        FFN_LOAD(fb, 1) FFN_GEMM1(fa, 0) FFN_DMA(0) FFN_DMA(1) FFN_DMA(2) FFN_PIN3()
        FFN_LOAD(fa, 2) FFN_GEMM1(fb, 1) FFN_DMA(3) FFN_DMA(4) FFN_DMA(5) FFN_PIN3()
        FFN_LOAD(fb, 3) FFN_GEMM1(fa, 2) FFN_DMA(6) FFN_DMA(7) FFN_DMA(8) FFN_PIN3()
        FFN_LOAD(fa, 4) FFN_GEMM1(fb, 3) FFN_DMA(9) FFN_DMA(10) FFN_DMA(11) FFN_PIN3()   // fa <- first group of W2 fragments
        __builtin_amdgcn_sched_barrier(0);
        // ---- relu(acc / row scale + bias), split into two fp16 planes: the lane's 2 x 4 values of a row group are its eight
        //      k-slots of the second product's B fragment ----
        const float* aux = reinterpret_cast<const float*>(smem + st * STAGE_BYTES + (W1_FRAGS + W2_FRAGS) * FRAG);
        half8 hf[2][RG];                                         // [plane][row group]
        {
            f32x4 sc[2], bi[2];
#pragma unroll
            for (int h_ = 0; h_ < 2; ++h_) {
                sc[h_] = *reinterpret_cast<const f32x4*>(aux + 16 * h_ + 4 * fg);
                bi[h_] = *reinterpret_cast<const f32x4*>(aux + CH + 16 * h_ + 4 * fg);
            }
#pragma unroll
            for (int r_ = 0; r_ < RG; ++r_) {
                f32x4 v[2];
#pragma unroll
                for (int h_ = 0; h_ < 2; ++h_)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[h_][e] = fmaxf(fmaf(acc1[h_][r_][e], sc[h_][e], bi[h_][e]), 0.f);
                        hmax = fmaxf(hmax, v[h_][e]);          // (never NaN after the max with 0) checked once, after the loop
                    }
                split8(v[0], v[1], hf[0][r_], hf[1][r_]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- Y^T += W2[:, chunk] . H^T : 16 output groups x 2 row groups, independent accumulators; fragment 2 i + p of group g =
        //      plane p of output group 4 g + i ----
#define FFN_GEMM2(src, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int t_ = (g) * 4 + i_;                                                                          \
        _Pragma("unroll") for (int r_ = 0; r_ < RG; ++r_) {                                                   \
            acc2[t_][r_] = mfma16(src[2 * i_ + 1], hf[0][r_], acc2[t_][r_]);                                  \
            acc2[t_][r_] = mfma16(src[2 * i_], hf[1][r_], acc2[t_][r_]);                                      \
            acc2[t_][r_] = mfma16(src[2 * i_], hf[0][r_], acc2[t_][r_]);                                      \
        }                                                                                                     \
    }
        FFN_LOAD(fb, 5) FFN_GEMM2(fa, 0) FFN_DMA(12) FFN_DMA(13) FFN_DMA(14) FFN_PIN3()
        FFN_LOAD(fa, 6) FFN_GEMM2(fb, 1) FFN_DMA(15) FFN_PIN1()
        FFN_LOAD(fb, 7) FFN_GEMM2(fa, 2) FFN_PIN0()
        FFN_GEMM2(fb, 3)
#undef FFN_DMA
#undef FFN_LOAD
#undef FFN_PIN3
#undef FFN_PIN1
#undef FFN_PIN0
#undef FFN_GEMM1
#undef FFN_GEMM2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of the next stage has landed
        __syncthreads();                                     // ... and everybody's; nobody still reads this stage
    }

    range_bad |= !(hmax <= 65504.f);                         // beyond fp16: flagged, never a silent wrong result
    // ---- epilogue: Y^T (row of X on the lane, output feature in the registers) -> row-major through LDS ----
    float* stg = reinterpret_cast<float*>(smem);             // [128][256] fp32; 16-byte chunk c of row r at chunk c ^ (r & 7)
#pragma unroll
    for (int r_ = 0; r_ < RG; ++r_) {
        const int lr = wave * WR + 16 * r_ + fn;
        float* mine = stg + lr * D;
#pragma unroll
        for (int t = 0; t < D / 16; ++t) {
            const int chunk = 4 * t + fg;                    // features 16 t + 4 g .. + 3
            *reinterpret_cast<f32x4*>(mine + ((chunk ^ (lr & 7)) << 2)) = acc2[t][r_];
        }
    }
    // The residual rows of the row pass below (row 4 g + rsel of the wave's 32, chunks sub + 16 k) are all requested HERE, in
    // the accumulators' registers: one HBM / L2 latency, under the barrier and the first staged reads, instead of one in front
    // of every pair of row groups.
    const int sub = lane & 15, rsel = lane >> 4;
    f32x4 xres[4 * RG][4];
#pragma unroll
    for (int g = 0; g < 4 * RG; ++g) {
        const long m = tile0 + wave * WR + 4 * g + rsel;
        const long mc = m < p.M ? m : p.M - 1;                  // tail rows recompute the last row, never stored
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (PLAIN) xres[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            else xres[g][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)mc * p.ldx + (sub + 16 * k) * 4);
        }
    }
    __syncthreads();
    // Row pass: FOUR rows per wave-instruction, 16 lanes per row, each lane four 16-byte column chunks (sub, sub + 16, ...):
    // the two LayerNorm reductions run over 16 lanes with four DPP steps each (quad swaps + the two row mirrors) instead of
    // six cross-lane permutes over the whole wave, and eight such groups per wave are independent chains the scheduler can
    // overlap (one row per instruction with a 64-lane butterfly measured 21 % of the kernel).
    f32x4 s2[4], b2[4], ga[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = (sub + 16 * k) * 4;
        s2[k] = *reinterpret_cast<const f32x4*>(p.s2 + col);
        b2[k] = *reinterpret_cast<const f32x4*>(p.b2 + col);
        if constexpr (!PLAIN) {
            ga[k] = *reinterpret_cast<const f32x4*>(p.gamma + col);
            be[k] = *reinterpret_cast<const f32x4*>(p.beta + col);
        }
    }
    int bad = range_bad;
    // Four row groups at a time: their sixteen staged chunks are read in one batch (one LDS latency, not sixteen).
#pragma unroll
    for (int gh = 0; gh < RG; ++gh) {
        f32x4 v[4][4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int lr = wave * WR + 4 * (4 * gh + gi) + rsel;   // row inside the workgroup's tile
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ch = sub + 16 * k;
                v[gi][k] = *reinterpret_cast<const f32x4*>(stg + lr * D + ((ch ^ (lr & 7)) << 2));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int g = 4 * gh + gi;
            const long m = tile0 + wave * WR + 4 * g + rsel;
            if constexpr (PLAIN) {
                const float lo = p.relu_out ? 0.f : -INFINITY;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x4 o = v[gi][k] * s2[k] + b2[k];
                    bad |= !(fabsf(o[0]) <= 3.4e38f) | !(fabsf(o[1]) <= 3.4e38f) | !(fabsf(o[2]) <= 3.4e38f) | !(fabsf(o[3]) <= 3.4e38f);
                    o = f32x4{fmaxf(o[0], lo), fmaxf(o[1], lo), fmaxf(o[2], lo), fmaxf(o[3], lo)};   // (the check sits in front)
                    if (m < p.M) *reinterpret_cast<f32x4*>(p.Y + (size_t)m * p.ldy + (sub + 16 * k) * 4) = o;
                }
                continue;
            }
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[gi][k] = v[gi][k] * s2[k] + b2[k] + xres[g][k];
                sum += (v[gi][k][0] + v[gi][k][1]) + (v[gi][k][2] + v[gi][k][3]);
            }
            const float mean = row16_sum(sum) * (1.f / D);
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[gi][k] = v[gi][k] - mean;
                q += (v[gi][k][0] * v[gi][k][0] + v[gi][k][1] * v[gi][k][1]) + (v[gi][k][2] * v[gi][k][2] + v[gi][k][3] * v[gi][k][3]);
            }
            const float rstd = rsqrtf(row16_sum(q) * (1.f / D) + p.eps);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 o = v[gi][k] * rstd * ga[k] + be[k];
                bad |= !(fabsf(o[0]) <= 3.4e38f) | !(fabsf(o[1]) <= 3.4e38f) | !(fabsf(o[2]) <= 3.4e38f) | !(fabsf(o[3]) <= 3.4e38f);
                if (m < p.M) *reinterpret_cast<f32x4*>(p.Y + (size_t)m * p.ldy + (sub + 16 * k) * 4) = o;
            }
        }
    }
    if (bad && p.flag) atomicOr(p.flag, 1);                  // an activation left fp16's range (gemm_f16x3.hip contract)
}

// Fragment-linear weight image.  Per chunk c of 32 hidden units, 65 fragments of 1 KB; element j (0..7) of lane l = (m, kg) =
// (l & 15, l >> 4):
//   f = 4 s + 2 Hh + p   (s = 0..7, Hh = 0..1)   : plane p of W1s[32 c + 16 Hh + m][32 s + 8 kg + j]
//   f = 32 + 2 t + p     (t = 0..15)             : plane p of W2s[16 t + m][32 c + 16 (j >> 2) + 4 kg + (j & 3)]
//   f = 64                                       : floats 0..31 = 1 / (row scale of W1s) of the chunk, 32..63 = b1 of the chunk
// (W1s / W2s = the row-scaled planes of gom_split_f16x2).
__global__ __launch_bounds__(256) void ffn_image_kernel(const unsigned short* __restrict__ p1, long ps1, int ld1,
                                                        const float* __restrict__ inv1, const float* __restrict__ b1,
                                                        const unsigned short* __restrict__ p2, long ps2, int ld2, int F,
                                                        unsigned short* __restrict__ img, int perm) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // one fp16 element of the image
    const long total = (long)(F / CH) * STAGE_FRAGS * 512;
    if (i >= total) return;
    const int e = (int)(i % 512), f = (int)((i / 512) % STAGE_FRAGS), c = (int)(i / (512L * STAGE_FRAGS));
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    if (f < W1_FRAGS) {
        const int s = f >> 2, hh = (f >> 1) & 1, pl = f & 1;
        // perm: the block's input arrives in ACCUMULATOR order (dec_tail.hip: the previous block's Y^T registers, split in place):
        // k-slot j of lane group kg at k-step s <-> input feature 32 s + 16 (j >> 2) + 4 kg + (j & 3)
        const int k = perm ? 32 * s + 16 * (j >> 2) + 4 * kg + (j & 3) : 32 * s + 8 * kg + j;
        img[i] = p1[pl * ps1 + (size_t)(CH * c + 16 * hh + m) * ld1 + k];
    } else if (f < W1_FRAGS + W2_FRAGS) {
        const int id = f - W1_FRAGS, t = id >> 1, pl = id & 1;
        img[i] = p2[pl * ps2 + (size_t)(16 * t + m) * ld2 + CH * c + 16 * (j >> 2) + 4 * kg + (j & 3)];
    } else {
        const int fi = e >> 1;                                // float index inside the fragment (two fp16 slots per float)
        float v = 0.f;
        if (fi < CH) v = inv1[CH * c + fi];
        else if (fi < 2 * CH) v = b1 ? b1[CH * c + fi - CH] : 0.f;
        const unsigned bits = __builtin_bit_cast(unsigned, v);
        img[i] = (unsigned short)((e & 1) ? (bits >> 16) : (bits & 0xffffu));
    }
}

}  // namespace

static int g_ffn_half_tail = 1;
static int g_ffn_dev_cus[16] = {0};                      // compute units per device (cached per device id)
static int g_ffn_stream_cus = 0;                         // > 0: the launches run on a CU-masked stream with this many CUs

// compute units the launch can use: the caller's figure for a CU-masked stream, else the device's
static int ffn_effective_cus() {
    if (g_ffn_stream_cus > 0) return g_ffn_stream_cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (g_ffn_dev_cus[dev] <= 0) {
        int n = 0;
        g_ffn_dev_cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return g_ffn_dev_cus[dev];
}
/* [host] the number of compute units the fused-FFN launches can run on when their stream is CU-masked (GoMatching.reserve_tracker_cus:
 * the detector on 224 of 256); 0 = the whole device.  Only the tail-round heuristic reads it: same bits either way. */
extern "C" int gom_ffn_set_stream_cus(int cus) {
    g_ffn_stream_cus = cus > 0 ? cus : 0;
    return GOM_OK;
}
/* [host] 1 (default): the partly filled last round of a long launch runs as half-height tiles; 0: one launch of 128-row tiles. */
extern "C" int gom_ffn_set_half_tail(int on) {
    g_ffn_half_tail = on ? 1 : 0;
    return GOM_OK;
}

extern "C" long gom_ffn_fused_image_bytes(int d_model, int d_hidden) {
    if (d_model != D || d_hidden <= 0 || (d_hidden % CH) != 0) return -1;
    return (long)(d_hidden / CH) * STAGE_BYTES;
}

static int ffn_image(const void* w1_planes, long w1_plane_stride, int ld1, const float* w1_inv_scale, const float* b1,
                     const void* w2_planes, long w2_plane_stride, int ld2, int d_model, int d_hidden, void* image, long image_bytes,
                     int perm, void* stream) {
    GOM_CHECK_ARG(w1_planes && w1_inv_scale && w2_planes && image);
    GOM_CHECK_ARG(d_model == D && d_hidden > 0 && (d_hidden % CH) == 0 && ld1 >= D && ld2 >= d_hidden);
    GOM_CHECK_ARG(image_bytes >= gom_ffn_fused_image_bytes(d_model, d_hidden));
    const long total = (long)(d_hidden / CH) * STAGE_FRAGS * 512;
    hipLaunchKernelGGL(ffn_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)w1_planes, w1_plane_stride, ld1, w1_inv_scale, b1,
                       (const unsigned short*)w2_planes, w2_plane_stride, ld2, d_hidden, (unsigned short*)image, perm);
    return gom_launch_status();
}

extern "C" int gom_ffn_fused_image(const void* w1_planes, long w1_plane_stride, int ld1, const float* w1_inv_scale,
                                   const float* b1, const void* w2_planes, long w2_plane_stride, int ld2, int d_model,
                                   int d_hidden, void* image, long image_bytes, void* stream) {
    return ffn_image(w1_planes, w1_plane_stride, ld1, w1_inv_scale, b1, w2_planes, w2_plane_stride, ld2, d_model, d_hidden, image,
                     image_bytes, 0, stream);
}

/* the same image for a block whose INPUT arrives in accumulator order (a block chained behind another one in dec_tail.hip) */
extern "C" int gom_ffn_fused_image_acc_order(const void* w1_planes, long w1_plane_stride, int ld1, const float* w1_inv_scale,
                                             const float* b1, const void* w2_planes, long w2_plane_stride, int ld2, int d_model,
                                             int d_hidden, void* image, long image_bytes, void* stream) {
    return ffn_image(w1_planes, w1_plane_stride, ld1, w1_inv_scale, b1, w2_planes, w2_plane_stride, ld2, d_model, d_hidden, image,
                     image_bytes, 1, stream);
}

extern "C" int gom_ffn_fused_ln_f32(const float* X, int ldx, const void* image, const float* w2_inv_scale, const float* b2,
                                    const float* gamma, const float* beta, float eps, float* Y, int ldy, int M, int d_model,
                                    int d_hidden, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && w2_inv_scale && b2 && gamma && beta && Y);
    GOM_CHECK_ARG(M >= 0 && d_model == D && d_hidden > 0 && (d_hidden % CH) == 0);
    GOM_CHECK_ARG(ldx >= D && ldy >= D && (ldx % 4) == 0 && (ldy % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)image % 16) == 0);
    if (M == 0) return GOM_OK;
    FfnArgs a{};
    a.X = X; a.img = (const unsigned char*)image; a.s2 = w2_inv_scale; a.b2 = b2; a.gamma = gamma; a.beta = beta; a.Y = Y;
    a.flag = flag; a.eps = eps; a.ldx = ldx; a.ldy = ldy; a.M = M; a.chunks = d_hidden / CH;
    const int tiles = cdiv(M, BM);
    a.stagger = tiles >= 1024 ? 8 : 0;                       // >= 4 rounds of workgroups
    // (the attribute is per DEVICE: set on every launch -- a process-wide flag would miss a second GPU; it costs ~1 us)
    hipError_t e = hipFuncSetAttribute((const void*)ffn_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)ffn_fused_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    // long launches whose last round of one-per-CU tiles is less than half full: that round as half-height tiles (same bits)
    const int cus = ffn_effective_cus();
    const int rem = tiles % cus;
    if (g_ffn_half_tail && tiles >= 4 * cus && rem > 0 && 2 * rem <= cus) {
        const int full = tiles - rem;
        hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3((unsigned)full), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
        a.row_base = (long)full * BM;
        a.stagger = 0;
        hipLaunchKernelGGL((ffn_fused_kernel<false, 1>), dim3((unsigned)cdiv(M - a.row_base, BM / 2)), dim3(256), LDS_BYTES,
                           (hipStream_t)stream, a);
        return gom_launch_status();
    }
    hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3((unsigned)tiles), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return gom_launch_status();
}

extern "C" int gom_mlp2_fused_f32(const float* X, int ldx, const void* image, const float* w2_inv_scale, const float* b2,
                                  int relu_out, float* Y, int ldy, int M, int d_model, int d_hidden, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && w2_inv_scale && b2 && Y);
    GOM_CHECK_ARG(M >= 0 && d_model == D && d_hidden > 0 && (d_hidden % CH) == 0);
    GOM_CHECK_ARG(ldx >= D && ldy >= D && (ldx % 4) == 0 && (ldy % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)image % 16) == 0 &&
                  ((uintptr_t)w2_inv_scale % 16) == 0 && ((uintptr_t)b2 % 16) == 0);
    if (M == 0) return GOM_OK;
    FfnArgs a{};
    a.X = X; a.img = (const unsigned char*)image; a.s2 = w2_inv_scale; a.b2 = b2; a.Y = Y; a.flag = flag; a.ldx = ldx; a.ldy = ldy;
    a.M = M; a.chunks = d_hidden / CH; a.stagger = cdiv(M, BM) >= 1024 ? 8 : 0; a.relu_out = relu_out ? 1 : 0;
    hipError_t e = hipFuncSetAttribute((const void*)ffn_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    hipLaunchKernelGGL(ffn_fused_kernel<true>, dim3((unsigned)cdiv(M, BM)), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return gom_launch_status();
}
