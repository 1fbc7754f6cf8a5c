// The two self-attention blocks of a DeepSolo composite decoder layer (dec_attn.hip's contract, same arithmetic), SECOND form:
// 16-token waves on v_mfma_f32_16x16x32_f16, EIGHT per workgroup = two per SIMD sharing one weight ring (bneck2.hip's scheme).
//
//   intra:  tgt = norm_intra(tgt + out_proj(MHA(q = k = tgt + query_pos, v = tgt)))   over the 25 points of one query
//   inter:  tgt = norm_inter(tgt + out_proj(MHA(q = k = v = tgt)))                    over the nq queries of one (frame, point)
//
// (/root/reference/third_party/adet/layers/deformable_transformer.py:386-404; nn.MultiheadAttention, 8 heads of 32, eval mode).
// Why: the first form runs one 32-token wave per SIMD, and its launch is one workgroup's serial chain -- 48 MFMAs per weight
// stage (1.5k cycles) inside ~4.5k of stage epilogues, splits, softmax and barriers (tools/exp/dec_attn_clock.py: 35 % of the
// matrix pipe); vector work does not hide inside a wave's own MFMA stream (LAB_NOTES, rounds 5-6), a second wave on the SIMD is what
// hides it (tools/exp/lds_rate.py: 94-98 % of the pipe with two waves against 72-87 % with one).  So:
//
//   * a wave owns 16 token slots as B / A operand fragments of the 16x16x32 shape (64 VGPRs), the eight heads' V resp. O^T in 64
//     more, the 256 out_proj columns in 64 accumulators: <= 256 registers, two waves per SIMD;
//   * an attention group spans waves -- intra: a PAIR of waves (tokens 0..15 | 16..G-1) = one query's points, four pairs per
//     workgroup; inter: all eight waves (ceil(G / 8) tokens each) -- which exchange a head's K and V fragments through 32 KB of LDS:
//     K as the A operand of S^T = K Q^T (one fragment per wave and plane), V as the A operand of O^T = V^T P^T, whose k index runs
//     over the KEYS of two waves: each of the two writes its half of every lane's 16 bytes.  All layout changes are in-lane:
//         q, k TRANSPOSED (D[feature][token]: lane (token, rg) holds features 16 Hh + 4 rg + e)  -> operand k index (kg, j) =
//             feature (j < 4 ? 4 kg + j : 16 + 4 kg + j - 4): the same permutation on Q and K;
//         v STRAIGHT (D[token][feature]: lane (feature, rg) holds tokens 4 rg + e) -> the keys (j < 4: wave 2 t, j >= 4: wave 2 t + 1)
//             4 kg + (j & 3) of k-step t: exactly what S^T's accumulators give P^T (lane (query, rg): keys 4 rg + e of each block);
//         O^T (lane (query, rg): d = 16 Hh + 4 rg + e) = out_proj's B operand in the same permuted k order, baked into its image;
//   * weights: the first form's stages (32 output columns x K = 256 = 32 fragments + 1 of (1 / row scale | bias)) in the 16x16x32
//     fragment order, three-slot LDS ring by MUBUF LDS-DMA, four pieces per wave and stage (wave 0: five);
//   * residual + LayerNorm in registers (lane = token: in-lane sums + two half-row exchanges), 64-byte pieces per token and store;
//   * RAW (inter): the cross attention's offsets | logits product on (output + query_pos) behind the block, twelve more stages.
// Plane products: w-lo x-hi, w-hi x-lo, w-hi x-hi per 32-wide k-step.  Range contract and *flag as gom_dec_attn_f32.
#include "common.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 256, NH = 8;
constexpr int FRAG = 1024;
constexpr int W_FRAGS = 32;
constexpr int CHUNK_FRAGS = W_FRAGS + 1;
constexpr int CHUNK_BYTES = CHUNK_FRAGS * FRAG;          // 33 KB
constexpr int STAGES = 3 * NH + NH;
constexpr int RAW_STAGES = 12;
constexpr int SLOTS = 3;
constexpr int RING_BYTES = SLOTS * CHUNK_BYTES;
constexpr int WAVES = 8;
constexpr int XCH_BYTES = 32 * FRAG;                     // K: [block 8][plane 2], V: [k-step 4][Hh 2][plane 2] fragments
constexpr int VEC_BYTES = 4 * FRAG;                      // out_proj's 1 / row scale | bias | gamma | beta, 256 floats each
constexpr int LDS_BYTES = RING_BYTES + XCH_BYTES + VEC_BYTES;
constexpr int IMAGE_BYTES = VEC_BYTES + STAGES * CHUNK_BYTES;            // the vectors lead, the stages follow
constexpr int RAW_IMAGE_BYTES = VEC_BYTES + (STAGES + RAW_STAGES) * CHUNK_BYTES;

struct DecArgs2 {
    const float* X;
    const float* P;
    const unsigned char* img;
    float* Y;
    int* flag;
    float eps, scale;
    int ldx, ldp, ldy;
    int groups, G, per_wave, inner;
    const float* P2;
    float* RAWO;
    int ldp2, ldraw;
};

__device__ __forceinline__ f32x4 mfma16(const half8 a, const half8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned frag_off, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)lane_off, (int)frag_off, 0, 0);
}

// -DA2_STAMPS (tools/dec_attn2_variants.py only): s_memtime between the phases of every wave, summed per kind into a buffer set by
// gom_dec_attn2_set_stamps -- [workgroup][wave][8] cycles: prologue, stage products, stage epilogues (+ exchange writes, stores),
// end-of-stage waits + barriers, attention, the intra form's row reload, residual + LayerNorm, total
#ifdef A2_STAMPS
__device__ unsigned long long* g_a2_stamps = nullptr;
__device__ __forceinline__ unsigned long long a2_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define A2_T(k)                                                        \
    {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                             \
        const unsigned long long n_ = a2_clock();                      \
        a2_b[k] += n_ - a2_last;                                       \
        a2_last = n_;                                                  \
        __builtin_amdgcn_sched_barrier(0);                             \
    }
#else
#define A2_T(k)
#endif

// exp2(x) where the lane's bit of `mask` (wave-uniform, in SGPRs) is set, else 0.  One asm block: the compiler's hazard recognizer
// does not look inside inline asm, and a VALU instruction must not read a transcendental's result in the very next slot (s_nop).
__device__ __forceinline__ float exp2_if(const float x, const unsigned long long mask) {
    float r;
    asm("v_exp_f32 %0, %1\n\ts_nop 0\n\tv_cndmask_b32 %0, 0, %0, %2" : "=&v"(r) : "v"(x), "s"(mask));
    return r;
}

// four fp32 values -> their fp16 planes: {hi01, hi23} and {lo01, lo23}
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
    unsigned h0, l0, h1, l1;
    gom_split2_f16(v[0], v[1], h0, l0);
    gom_split2_f16(v[2], v[3], h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}

// A wave's 16 rows x 256 fp32 in FRAGMENT order: consume(s, a, b) receives, for every 32-wide k-step s, lane (n = lane & 15, kg =
// lane >> 4)'s floats 32 s + 8 kg .. + 3 (a) and .. + 4 .. + 7 (b) of row n.  Whole-line loads (a wave-instruction = 64 floats of four
// rows: 8 lines, where the fragment layout straight from memory would touch 16 -- the texture-address unit pays per line, LAB_NOTES) and
// a layout change in a wave-private 4 KB scratch, 16-byte pieces XOR-swizzled by the row (common.h gom_rows_to_fragments for one
// 16-row group).  ALL loads are issued before the first exchange: their latency is paid once.  ADD: the rows are row_a + row_b.
template <typename FA>
__device__ __forceinline__ void rows16_load(FA row_a, int lane, f32x4 (&v)[4][4]) {
    const int pc = lane & 15, r0 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* ra = row_a(r0 + 4 * i) + pc * 4;
#pragma unroll
        for (int part = 0; part < 4; ++part) v[part][i] = *reinterpret_cast<const f32x4*>(ra + part * 64);
    }
}
template <typename FC>
__device__ __forceinline__ void rows16_exchange(const f32x4 (&v)[4][4], float* scratch, int lane, FC consume) {
    const int pc = lane & 15, r0 = lane >> 4;
    const int fn = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int part = 0; part < 4; ++part) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * i;
            *reinterpret_cast<f32x4*>(scratch + r * 64 + ((pc ^ (r & 15)) << 2)) = v[part][i];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int p0 = 8 * s + 2 * fg;
            const f32x4 a = *reinterpret_cast<const f32x4*>(scratch + fn * 64 + ((p0 ^ fn) << 2));
            const f32x4 b = *reinterpret_cast<const f32x4*>(scratch + fn * 64 + (((p0 + 1) ^ fn) << 2));
            consume(2 * part + s, a, b);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// the rows (ADD: row_a + row_b), one 64-float part at a time (measured against two parts in flight, and against all sixteen loads up
// front: 9.7k cycles for the intra form's reload against 13.6k -- the texture-address unit, not the latency, bounds eight waves
// loading at once)
template <bool ADD, typename FA, typename FB, typename FH, typename FC>
__device__ __forceinline__ void rows16_stream(FA row_a, FB row_b, float* scratch, int lane, FH after_loads, FC consume) {
    const int pc = lane & 15, r0 = lane >> 4;
    const int fn = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * i;
            v[i] = *reinterpret_cast<const f32x4*>(row_a(r) + part * 64 + pc * 4);
            if constexpr (ADD) v[i] += *reinterpret_cast<const f32x4*>(row_b(r) + part * 64 + pc * 4);
        }
        if (part == 0) {
            __builtin_amdgcn_sched_barrier(0);
            after_loads();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * i;
            *reinterpret_cast<f32x4*>(scratch + r * 64 + ((pc ^ (r & 15)) << 2)) = v[i];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int p0 = 8 * s + 2 * fg;
            const f32x4 a = *reinterpret_cast<const f32x4*>(scratch + fn * 64 + ((p0 ^ fn) << 2));
            const f32x4 b = *reinterpret_cast<const f32x4*>(scratch + fn * 64 + (((p0 + 1) ^ fn) << 2));
            consume(2 * part + s, a, b);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ... as the two fp16 planes of the wave's MFMA operand fragments, xf[plane][k-step]
template <bool ADD, typename FA, typename FB, typename FH>
__device__ __forceinline__ void rows16_to_fragments(FA row_a, FB row_b, float* scratch, int lane, half8 (&xf)[2][8], float& amax,
                                                    FH after_loads) {
    rows16_stream<ADD>(row_a, row_b, scratch, lane, after_loads, [&](const int s, const f32x4 a, const f32x4 b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(a[e]), fabsf(b[e])));
        gom_split8_f16(a, b, xf[0][s], xf[1][s]);
    });
    asm volatile("" : "+v"(amax));
}

// one weight stage: 32 fragments in eight groups of four (k-step g), group g + 1 read while the MFMAs of group g issue (four-fragment
// groups: the two-deep register pipeline costs 32 VGPRs, not 64); this wave's four pieces of stage i + 2 are requested under every
// other group
#define A2_LOAD(dst, g)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                          \
        dst[i_] = *reinterpret_cast<const half8*>(base + ((g) * 4 + i_) * FRAG);
#define A2_DMA(k) dma_fragment(rs_img, lane16, nsrc + (k) * WAVES * FRAG, ndst + (k) * WAVES * FRAG);
#define A2_PIN()                                          \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#define A2_PIN0()                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
#define A2_STAGE(MFMA)                                                                                        \
    {                                                                                                         \
        half8 fa[4], fb[4];                                                                                   \
        A2_LOAD(fa, 0)                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                    \
        A2_LOAD(fb, 1) MFMA(fa, 0) A2_DMA(0) A2_PIN()                                                         \
        A2_LOAD(fa, 2) MFMA(fb, 1) A2_PIN0()                                                                  \
        A2_LOAD(fb, 3) MFMA(fa, 2) A2_DMA(1) A2_PIN()                                                         \
        A2_LOAD(fa, 4) MFMA(fb, 3) A2_PIN0()                                                                  \
        A2_LOAD(fb, 5) MFMA(fa, 4) A2_DMA(2) A2_PIN()                                                         \
        A2_LOAD(fa, 6) MFMA(fb, 5) A2_PIN0()                                                                  \
        A2_LOAD(fb, 7) MFMA(fa, 6) A2_DMA(3) A2_PIN()                                                         \
        MFMA(fb, 7)                                                                                           \
        A2_T(1)                                                                                               \
    }
#define A2_STAGE_LAST(MFMA)                                                                                   \
    {                                                                                                         \
        half8 fa[4], fb[4];                                                                                   \
        A2_LOAD(fa, 0)                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                    \
        A2_LOAD(fb, 1) MFMA(fa, 0) A2_PIN0()                                                                  \
        A2_LOAD(fa, 2) MFMA(fb, 1) A2_PIN0()                                                                  \
        A2_LOAD(fb, 3) MFMA(fa, 2) A2_PIN0()                                                                  \
        A2_LOAD(fa, 4) MFMA(fb, 3) A2_PIN0()                                                                  \
        A2_LOAD(fb, 5) MFMA(fa, 4) A2_PIN0()                                                                  \
        A2_LOAD(fa, 6) MFMA(fb, 5) A2_PIN0()                                                                  \
        A2_LOAD(fb, 7) MFMA(fa, 6) A2_PIN0()                                                                  \
        MFMA(fb, 7)                                                                                           \
        A2_T(1)                                                                                               \
    }
// fragment 2 Hh + p of a group = plane p of feature tile Hh at k-step g
// transposed: acc[Hh][feature 16 Hh + 4 rg + e][token] += W . X^T   (A = weight fragment, B = the rows)
#define A2_MFMA_T(src, g)                                                                                     \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc[h_] = mfma16(src[2 * h_ + 1], xf[0][g], acc[h_]);    \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc[h_] = mfma16(src[2 * h_], xf[1][g], acc[h_]);        \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc[h_] = mfma16(src[2 * h_], xf[0][g], acc[h_]);
// straight: acc[Hh][token 4 rg + e][feature 16 Hh + n] += X . W^T   (A = the rows, B = weight fragment)
#define A2_MFMA_S(src, g)                                                                                     \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc[h_] = mfma16(xf[0][g], src[2 * h_ + 1], acc[h_]);    \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc[h_] = mfma16(xf[1][g], src[2 * h_], acc[h_]);        \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) acc[h_] = mfma16(xf[0][g], src[2 * h_], acc[h_]);
// out_proj stage of a head: fragment 2 t + p = plane p of output tile t; group g = tiles 2 g, 2 g + 1
#define A2_MFMA_O(src, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) yacc[(g) * 2 + i_] = mfma16(src[2 * i_ + 1], o_hi, yacc[(g) * 2 + i_]); \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) yacc[(g) * 2 + i_] = mfma16(src[2 * i_], o_lo, yacc[(g) * 2 + i_]);     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) yacc[(g) * 2 + i_] = mfma16(src[2 * i_], o_hi, yacc[(g) * 2 + i_]);

template <bool INTER, bool RAW = false>
__global__ __launch_bounds__(512, 1) void dec_attn2_kernel(const DecArgs2 p) {
    constexpr int NST = STAGES + (RAW ? RAW_STAGES : 0);
    constexpr int NB = INTER ? 8 : 2;                        // key blocks (waves) of an attention group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xch_k = smem + RING_BYTES;
    unsigned char* xch_v = xch_k + 16 * FRAG;
    const float* vecs = reinterpret_cast<const float*>(smem + RING_BYTES + XCH_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fn = lane & 15, fg = lane >> 4;
    const int b0 = INTER ? 0 : (wave & ~1);                  // first key block of this wave's attention group
#ifdef A2_STAMPS
    unsigned long long a2_b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long a2_t0 = a2_clock();
    unsigned long long a2_last = a2_t0;
#endif
    const int t0 = b0 >> 1;                                  // and its first V k-step

    // ---- token slot r (0..15) of this wave -> its row; slots beyond the group's tokens recompute token 0 (masked as keys, never stored)
    long gbase, gstep;                                       // row of token tq = gbase + tq * gstep
    int first, mine;                                         // this wave's first token and its token count
    int n_full, tok_full, tok_last;                          // key blocks 0 .. n_full - 1 hold tok_full tokens, block n_full tok_last, the rest none
    if constexpr (!INTER) {
        const long gi = (long)blockIdx.x * 4 + (wave >> 1);
        const long g = gi < p.groups ? gi : p.groups - 1;
        gbase = g * p.G;
        gstep = 1;
        first = (wave & 1) * 16;
        const int left = p.G - first;
        mine = gi < p.groups ? (left < 0 ? 0 : (left < 16 ? left : 16)) : 0;
        n_full = 1;
        tok_full = p.G < 16 ? p.G : 16;
        tok_last = p.G > 16 ? p.G - 16 : 0;
    } else {
        const long gi = blockIdx.x;
        const long b = gi / p.inner, pp = gi % p.inner;
        gbase = b * p.G * p.inner + pp;
        gstep = p.inner;
        first = wave * p.per_wave;
        const int left = p.G - first;
        mine = left < 0 ? 0 : (left < p.per_wave ? left : p.per_wave);
        n_full = p.G / p.per_wave;
        tok_full = p.per_wave;
        tok_last = p.G - n_full * p.per_wave;
    }
    unsigned long long m_full[4], m_last[4];                 // lanes whose key slot 4 fg + e is a token, per kind of block
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        m_full[e] = __builtin_amdgcn_ballot_w64(4 * fg + e < tok_full);
        m_last[e] = __builtin_amdgcn_ballot_w64(4 * fg + e < tok_last);
    }
    auto slot_row = [&](int r) -> long { return gbase + (r < mine ? (long)(first + r) : 0L) * gstep; };
    const bool valid = fn < mine;
    const long row = slot_row(fn);

    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, VEC_BYTES + NST * CHUNK_BYTES, 0x00020000);
    constexpr unsigned OOB = 0x7FFF0000u;
    const unsigned lane16 = lane * 16;

    float amax = 0.f, chk = 0.f;
    half8 xf[2][8];
    auto xrow = [&](int r) { return p.X + (size_t)slot_row(r) * p.ldx; };
    auto prow = [&](int r) { return p.P + (size_t)slot_row(r) * p.ldp; };
    unsigned pf[1];
    {
        // the ring's first two stages and the epilogue vectors are requested behind the first loads, slot 2 is the scratch
        float* scratch = reinterpret_cast<float*>(smem + 2 * CHUNK_BYTES) + wave * (16 * 64);
        rows16_to_fragments<false>(xrow, xrow, scratch, lane, xf, amax, [&]() {
            for (int f = wave; f < 2 * CHUNK_FRAGS; f += WAVES) dma_fragment(rs_img, lane16, VEC_BYTES + f * FRAG, smem + f * FRAG);
            if (wave < 4) dma_fragment(rs_img, lane16, wave * FRAG, smem + RING_BYTES + XCH_BYTES + wave * FRAG);
            // the whole image towards this XCD's L2 (common.h gom_prefetch_image): between the layers of a step it (1.1 - 1.5 MB, last
            // read a step ago) is in HBM, and the ring's two stages of lookahead do not cover a miss per stage (tools/
            // dec_attn2_variants.py, a 768 MB fill in front of every launch: inter + raw 76 -> 99 us, 88 with this)
            gom_prefetch_image(p.img, (unsigned)(VEC_BYTES + NST * CHUNK_BYTES), tid, 512, pf);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gom_prefetch_done(pf);
    __syncthreads();
    A2_T(0)

#define A2_STAGE_VARS(i)                                                                                      \
    const unsigned char* base = smem + ((i) % SLOTS) * CHUNK_BYTES + lane * 16;                               \
    const float* aux = reinterpret_cast<const float*>(smem + ((i) % SLOTS) * CHUNK_BYTES + W_FRAGS * FRAG);   \
    const unsigned nsrc = (i) + 2 < NST ? (unsigned)VEC_BYTES + (unsigned)((i) + 2) * CHUNK_BYTES + wave * FRAG : OOB; \
    unsigned char* ndst = smem + (((i) + 2) % SLOTS) * CHUNK_BYTES + wave * FRAG;                             \
    if (wave == 0) dma_fragment(rs_img, lane16, (i) + 2 < NST ? (unsigned)VEC_BYTES + (unsigned)((i) + 2) * CHUNK_BYTES + W_FRAGS * FRAG : OOB, \
                                smem + (((i) + 2) % SLOTS) * CHUNK_BYTES + W_FRAGS * FRAG);                   \
    __builtin_amdgcn_sched_barrier(0);
    // end of a stage: everything older than this stage's four (wave 0: five) requests has landed (= stage i + 1, requested a stage
    // ago; loads return in issue order); the LDS writes of the exchange are covered by the barrier's fence
#define A2_STAGE_END()                                                                                        \
    A2_T(2)                                                                                                   \
    if (wave == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                           \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                     \
    __syncthreads();                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    A2_T(3)
#define A2_STAGE_END_ALL()                                                                                    \
    A2_T(2)                                                                                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    __syncthreads();                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    A2_T(3)

    // transposed stage epilogue: value = acc * (1 / row scale) + bias of features 16 Hh + 4 fg + e
    auto finish_t = [&](f32x4 (&acc)[2], const float* aux) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + 16 * h + 4 * fg);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(aux + 32 + 16 * h + 4 * fg);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[h][e] = fmaf(acc[h][e], sc[e], bi[e]);
                amax = fmaxf(amax, fabsf(acc[h][e]));
            }
        }
        asm volatile("" : "+v"(amax));
    };
    // straight stage epilogue: the lane is the feature
    auto finish_s = [&](f32x4 (&acc)[2], const float* aux) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float sc = aux[16 * h + fn], bi = aux[32 + 16 * h + fn];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[h][e] = fmaf(acc[h][e], sc, bi);
                amax = fmaxf(amax, fabsf(acc[h][e]));
            }
        }
        asm volatile("" : "+v"(amax));
    };
    // this wave's K fragment (A operand of S^T: lane (key, kg), the head's features in accumulator order) -> block `wave`
    auto put_k = [&](const f32x4 (&acc)[2]) {
        half8 hi, lo;
        gom_split8_f16(acc[0], acc[1], hi, lo);
        *reinterpret_cast<half8*>(xch_k + (wave * 2 + 0) * FRAG + lane * 16) = hi;
        *reinterpret_cast<half8*>(xch_k + (wave * 2 + 1) * FRAG + lane * 16) = lo;
    };
    // this wave's half of the V fragments of k-step wave >> 1 (A operand of O^T: lane (d, kg), keys 4 kg + e of this wave)
    auto put_v = [&](const u32x2 hi0, const u32x2 lo0, const u32x2 hi1, const u32x2 lo1) {
        unsigned char* dst = xch_v + ((wave >> 1) * 4) * FRAG + lane * 16 + 8 * (wave & 1);
        *reinterpret_cast<u32x2*>(dst + 0 * FRAG) = hi0;
        *reinterpret_cast<u32x2*>(dst + 1 * FRAG) = lo0;
        *reinterpret_cast<u32x2*>(dst + 2 * FRAG) = hi1;
        *reinterpret_cast<u32x2*>(dst + 3 * FRAG) = lo1;
    };

    // vo[h]: sweep 1 of the intra form parks V of head h here ({hi, lo} halves of tile Hh in vo[h][Hh]); after the head's attention
    // it holds O^T of the head as out_proj's B operand (vo[h][plane])
    u32x4 vo[NH][2];

    // head h's attention for this wave's 16 queries: q in registers, K / V of the group's waves in the exchange
    auto attend = [&](const half8 q_hi, const half8 q_lo, const int h) {
        f32x4 s[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) s[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int KB = NB < 4 ? NB : 4;                  // key blocks per batch of fragment reads
#pragma unroll
        for (int bb = 0; bb < NB; bb += KB) {
            half8 k_hi[KB], k_lo[KB];
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                k_hi[b] = *reinterpret_cast<const half8*>(xch_k + ((b0 + bb + b) * 2 + 0) * FRAG + lane * 16);
                k_lo[b] = *reinterpret_cast<const half8*>(xch_k + ((b0 + bb + b) * 2 + 1) * FRAG + lane * 16);
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) s[bb + b] = mfma16(k_lo[b], q_hi, s[bb + b]);
#pragma unroll
            for (int b = 0; b < KB; ++b) s[bb + b] = mfma16(k_hi[b], q_lo, s[bb + b]);
#pragma unroll
            for (int b = 0; b < KB; ++b) s[bb + b] = mfma16(k_hi[b], q_hi, s[bb + b]);
        }
        // the first k-step's V fragments: on their way under the softmax
        half8 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const half8*>(xch_v + (t0 * 4 + i) * FRAG + lane * 16);
        // softmax over the keys: lane (query, fg) holds keys 4 fg + e of every block, already x scale x log2(e) (folded into q's
        // stage vectors by the image).  The maximum runs over EVERY slot: a padded slot recomputes a token of the group, so its score
        // is one of the valid ones; the probabilities of padded slots are cleared by lane masks made once per kernel.  (Reductions
        // as trees: the compiler keeps the order it is given, and 32 dependent fmaxf / adds are 32 issue latencies.)
        float mb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) mb[b] = fmaxf(fmaxf(s[b][0], s[b][1]), fmaxf(s[b][2], s[b][3]));
#pragma unroll
        for (int w = 1; w < NB; w *= 2)
#pragma unroll
            for (int b = 0; b + w < NB; b += 2 * w) mb[b] = fmaxf(mb[b], mb[b + w]);
        float mx = mb[0];
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long km = b < n_full ? m_full[e] : (b == n_full ? m_last[e] : 0ull);
                s[b][e] = exp2_if(s[b][e] - mx, km);
            }
            sb[b] = (s[b][0] + s[b][1]) + (s[b][2] + s[b][3]);
        }
#pragma unroll
        for (int w = 1; w < NB; w *= 2)
#pragma unroll
            for (int b = 0; b + w < NB; b += 2 * w) sb[b] += sb[b + w];
        float sum = sb[0];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < NB / 2; ++t) {
            half8 p_hi, p_lo;
            gom_split8_f16(s[2 * t], s[2 * t + 1], p_hi, p_lo);
            half8 vn[4];
            if (t + 1 < NB / 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vn[i] = *reinterpret_cast<const half8*>(xch_v + ((t0 + t + 1) * 4 + i) * FRAG + lane * 16);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) o[hh] = mfma16(v[2 * hh + 1], p_hi, o[hh]);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) o[hh] = mfma16(v[2 * hh], p_lo, o[hh]);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) o[hh] = mfma16(v[2 * hh], p_hi, o[hh]);
            if (t + 1 < NB / 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = vn[i];
            }
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[hh][e] *= inv;
                amax = fmaxf(amax, fabsf(o[hh][e]));
            }
        asm volatile("" : "+v"(amax));
        half8 o_hi, o_lo;
        gom_split8_f16(o[0], o[1], o_hi, o_lo);
        vo[h][0] = __builtin_bit_cast(u32x4, o_hi);
        vo[h][1] = __builtin_bit_cast(u32x4, o_lo);
    };

    // Head h's attention needs the K / V fragments of the stage(s) just ended and must be over before the NEXT head's k stage writes the
    // exchange again: it runs at the head of the interval of the stage that follows (the next head's q stage, or the first out_proj
    // stage).  (Measured and dropped: the waves 4..7 -- which share their SIMDs with the waves 0..3 -- running it BEHIND that stage's
    // products instead, so that one wave's softmax runs under the other's MFMAs: inter 127.6k -> 132.2k cycles.  A head's attention is
    // a LATENCY chain -- exchange reads, dependent MFMAs, two cross-lane reductions, exp, splits: ~4k cycles per wave whether or not
    // the SIMD's other wave is in it too -- and two of them overlap each other better than one overlaps 768 cycles of MFMAs.)
    constexpr bool early = true;
    half8 qp_hi, qp_lo;                                      // q of the head whose attention is pending
#define A2_ATTEND_EARLY(h)                                                                                    \
    if ((h) >= 0 && early) {                                                                                  \
        attend(qp_hi, qp_lo, (h));                                                                            \
        A2_T(4)                                                                                               \
    }
#define A2_ATTEND_LATE(h)                                                                                     \
    if ((h) >= 0 && !early) {                                                                                 \
        attend(qp_hi, qp_lo, (h));                                                                            \
        A2_T(4)                                                                                               \
    }
    if constexpr (!INTER) {
        // ---- sweep 1: V of every head (stages 0 .. 7), this wave's halves parked in vo[h] until the head's attention ----
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            A2_STAGE_VARS(h)
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            A2_STAGE(A2_MFMA_S)
            finish_s(acc, aux);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                u32x2 hi, lo;
                split4(acc[hh], hi, lo);
                vo[h][hh] = u32x4{hi[0], hi[1], lo[0], lo[1]};
            }
            A2_STAGE_END()
        }
        // ---- sweep 2: q | k of (tgt + query_pos) per head (stages 8 + 2 h, 9 + 2 h) ----
        __builtin_amdgcn_sched_barrier(0);
        {
            // (stage 8 sits in slot 2, stage 9 in slot 0; slot 1 held stage 7 and is free until stage 8 requests stage 10 into it)
            // only query_pos is loaded: tgt comes back from its own fragments (hi + lo: the 22-bit value every product of sweep 1
            // saw) -- half the lines for the texture-address unit, which bounds eight waves loading at once
            float* scratch = reinterpret_cast<float*>(smem + 1 * CHUNK_BYTES) + wave * (16 * 64);
            rows16_stream<false>(prow, prow, scratch, lane, [&]() {}, [&](const int s, const f32x4 a, const f32x4 b) {
                f32x4 xa, xb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xa[e] = ((float)xf[0][s][e] + (float)xf[1][s][e]) + a[e];
                    xb[e] = ((float)xf[0][s][4 + e] + (float)xf[1][s][4 + e]) + b[e];
                    amax = fmaxf(amax, fmaxf(fabsf(xa[e]), fabsf(xb[e])));
                }
                gom_split8_f16(xa, xb, xf[0][s], xf[1][s]);
            });
            asm volatile("" : "+v"(amax));
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        A2_T(5)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            {
                A2_ATTEND_EARLY(h - 1)
                A2_STAGE_VARS(NH + 2 * h)
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                A2_STAGE(A2_MFMA_T)
                A2_ATTEND_LATE(h - 1)
                finish_t(acc, aux);
                gom_split8_f16(acc[0], acc[1], qp_hi, qp_lo);
                A2_STAGE_END()
            }
            {
                A2_STAGE_VARS(NH + 2 * h + 1)
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                A2_STAGE(A2_MFMA_T)
                finish_t(acc, aux);
                put_k(acc);
                put_v(u32x2{vo[h][0][0], vo[h][0][1]}, u32x2{vo[h][0][2], vo[h][0][3]}, u32x2{vo[h][1][0], vo[h][1][1]},
                      u32x2{vo[h][1][2], vo[h][1][3]});
                A2_STAGE_END()
            }
        }
    } else {
        // ---- per head: q, k, v of tgt (stages 3 h, 3 h + 1, 3 h + 2) ----
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            {
                A2_ATTEND_EARLY(h - 1)
                A2_STAGE_VARS(3 * h)
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                A2_STAGE(A2_MFMA_T)
                A2_ATTEND_LATE(h - 1)
                finish_t(acc, aux);
                gom_split8_f16(acc[0], acc[1], qp_hi, qp_lo);
                A2_STAGE_END()
            }
            {
                A2_STAGE_VARS(3 * h + 1)
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                A2_STAGE(A2_MFMA_T)
                finish_t(acc, aux);
                put_k(acc);
                A2_STAGE_END()
            }
            {
                A2_STAGE_VARS(3 * h + 2)
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                A2_STAGE(A2_MFMA_S)
                finish_s(acc, aux);
                u32x2 hi0, lo0, hi1, lo1;
                split4(acc[0], hi0, lo0);
                split4(acc[1], hi1, lo1);
                put_v(hi0, lo0, hi1, lo1);
                A2_STAGE_END()
            }
        }
    }

    // ---- out_proj: Y^T[256 x tokens] += Wo[:, head h's features] . O_h^T, eight stages.  Its image orders the output rows so that
    // lane (token, fg) holds features 32 s + 8 fg .. + 7 of its token in yacc[2 s], yacc[2 s + 1] -- the FRAGMENT order of the rows:
    // the residual rows (and the RAW form's query_pos) then arrive as WHOLE lines through the exchange area, idle since the last head's
    // attention (8 lines per wave-instruction where the accumulator layout straight from memory touches 16); they are requested in
    // front of the last stage (the heads' fragments are dead by then: their registers) ----
    f32x4 yacc[D / 16];
#pragma unroll
    for (int t = 0; t < D / 16; ++t) yacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rv[4][4];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const half8 o_hi = __builtin_bit_cast(half8, vo[h][0]), o_lo = __builtin_bit_cast(half8, vo[h][1]);
        if (h == NH - 1) {
            rows16_load(xrow, lane, rv);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (RAW || h < NH - 2) {
            if (h == 0) { A2_ATTEND_EARLY(NH - 1) }
            A2_STAGE_VARS(3 * NH + h)
            (void)aux;
            A2_STAGE(A2_MFMA_O)
            if (h == 0) { A2_ATTEND_LATE(NH - 1) }
            A2_STAGE_END()
        } else {
            const unsigned char* base = smem + ((3 * NH + h) % SLOTS) * CHUNK_BYTES + lane * 16;
            A2_STAGE_LAST(A2_MFMA_O)
            if (h == NH - 2) { A2_STAGE_END_ALL() }
        }
    }

    // ---- residual + LayerNorm in registers (lane (token, fg): features 32 s + 8 fg .. + 7 in yacc[2 s], yacc[2 s + 1]) ----
    __builtin_amdgcn_sched_barrier(0);
    {
        const float* v_inv = vecs + 8 * fg;
        const float* v_bias = vecs + 256 + 8 * fg;
        const float* v_gamma = vecs + 512 + 8 * fg;
        const float* v_beta = vecs + 768 + 8 * fg;
        float* scratch = reinterpret_cast<float*>(xch_k) + wave * (16 * 64);
        float sum = 0.f;
        rows16_exchange(rv, scratch, lane, [&](const int s2, const f32x4 a, const f32x4 b) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(v_inv + 32 * s2 + 4 * i);
                const f32x4 bi = *reinterpret_cast<const f32x4*>(v_bias + 32 * s2 + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(yacc[2 * s2 + i][e], sc[e], bi[e]) + (i ? b[e] : a[e]);
                    yacc[2 * s2 + i][e] = v;
                    sum += v;
                }
            }
        });
        if constexpr (RAW) {
            __builtin_amdgcn_sched_barrier(0);
            rows16_load([&](int r) { return p.P2 + (size_t)slot_row(r) * p.ldp2; }, lane, rv);   // query_pos: under the statistics
            __builtin_amdgcn_sched_barrier(0);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.f / D);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < D / 16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                yacc[t][e] -= mean;
                sq += yacc[t][e] * yacc[t][e];
            }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.f / D) + p.eps);
        float* yr = p.Y + (size_t)row * p.ldy + 8 * fg;
        // (s2, a, b): the output's features 32 s2 + 8 fg .. + 7; RAW: a | b = query_pos there, and (output + query_pos) becomes k-step s2
        // of the NEXT product's B operand
        auto finish = [&](const int s2, const f32x4 a, const f32x4 b) {
            f32x4 x2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(v_gamma + 32 * s2 + 4 * i);
                const f32x4 be = *reinterpret_cast<const f32x4*>(v_beta + 32 * s2 + 4 * i);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = yacc[2 * s2 + i][e] * rstd * ga[e] + be[e];
                    chk = fmaf(o[e], 0.f, chk);
                }
                if (valid) *reinterpret_cast<f32x4*>(yr + 32 * s2 + 4 * i) = o;
                if constexpr (RAW) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x2[i][e] = o[e] + (i ? b[e] : a[e]);
                        amax = fmaxf(amax, fabsf(x2[i][e]));
                    }
                }
            }
            if constexpr (RAW) gom_split8_f16(x2[0], x2[1], xf[0][s2], xf[1][s2]);
        };
        if constexpr (RAW) {
            rows16_exchange(rv, scratch, lane, finish);
        } else {
#pragma unroll
            for (int s2 = 0; s2 < D / 32; ++s2) finish(s2, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        asm volatile("" : "+v"(chk), "+v"(amax));
    }
    A2_T(6)
    if constexpr (RAW) {
        // ---- raw = (Y + query_pos) Wraw^T + braw: twelve 32-column stages, transposed (lane = token); a stage's two stores are
        // issued at the START of the next one (then the oldest vector-memory operations of that stage: dec_attn.hip) ----
        __builtin_amdgcn_sched_barrier(0);
        float* ro = p.RAWO + (size_t)row * p.ldraw + 4 * fg;
        f32x4 pend[2];
#pragma unroll
        for (int c = 0; c < RAW_STAGES; ++c) {
            if (c > 0 && valid) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<f32x4*>(ro + 32 * (c - 1) + 16 * hh) = pend[hh];
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            if (c < RAW_STAGES - 2) {
                A2_STAGE_VARS(STAGES + c)
                A2_STAGE(A2_MFMA_T)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + 16 * hh + 4 * fg);
                    const f32x4 bi = *reinterpret_cast<const f32x4*>(aux + 32 + 16 * hh + 4 * fg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pend[hh][e] = fmaf(acc[hh][e], sc[e], bi[e]);
                        chk = fmaf(pend[hh][e], 0.f, chk);
                    }
                }
                asm volatile("" : "+v"(chk));
                A2_STAGE_END()
            } else {
                const unsigned char* base = smem + ((STAGES + c) % SLOTS) * CHUNK_BYTES + lane * 16;
                const float* aux = reinterpret_cast<const float*>(smem + ((STAGES + c) % SLOTS) * CHUNK_BYTES + W_FRAGS * FRAG);
                A2_STAGE_LAST(A2_MFMA_T)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + 16 * hh + 4 * fg);
                    const f32x4 bi = *reinterpret_cast<const f32x4*>(aux + 32 + 16 * hh + 4 * fg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pend[hh][e] = fmaf(acc[hh][e], sc[e], bi[e]);
                        chk = fmaf(pend[hh][e], 0.f, chk);
                    }
                }
                asm volatile("" : "+v"(chk));
                if (c == RAW_STAGES - 2) { A2_STAGE_END_ALL() }
            }
        }
        if (valid) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<f32x4*>(ro + 32 * (RAW_STAGES - 1) + 16 * hh) = pend[hh];
        }
    }
#ifdef A2_STAMPS
    A2_T(2)
    if (g_a2_stamps && lane == 0) {
        unsigned long long* o = g_a2_stamps + ((size_t)blockIdx.x * WAVES + wave) * 8;
#pragma unroll
        for (int i = 0; i < 7; ++i) o[i] = a2_b[i];
        o[7] = a2_last - a2_t0;
    }
#endif
    if ((!(amax <= 65504.f) || !(chk == 0.f)) && p.flag) atomicOr(p.flag, 1);
}

// Fragment-linear image of a block's weights for the 16x16x32 shape (row-scaled planes of gom_split_f16x2).  Bytes 0 .. 4095:
// out_proj's 1 / row scale | bias | gamma | beta (256 floats each).  Stages behind them, in dec_attn.hip's order -- intra: v_0 .. v_7,
// then (q_h, k_h); inter: (q_h, k_h, v_h); then the eight out_proj stages.  Element j of lane l = (m, kg) = (l & 15, l >> 4):
//   projection stage, rows row0 .. row0 + 31 of in_proj:  fragment 4 s + 2 Hh + p = plane p of Ws[row0 + 16 Hh + m][32 s + 8 kg + j];
//                                                         fragment 32: floats 0..31 = 1 / row scale, 32..63 = bias (q rows: both
//                                                         x 1 / sqrt(32) x log2(e))
//   out_proj stage hd: fragment 2 t + p = plane p of Wo[32 (t >> 1) + 8 (m >> 2) + 4 (t & 1) + (m & 3)][32 hd + 16 (j >> 2) + 4 kg + (j & 3)]
//       (columns: O^T's accumulator order; rows: so that a lane's accumulators of tiles 2 s, 2 s + 1 are features 32 s + 8 rg .. + 7)
__global__ __launch_bounds__(256) void dec_attn2_image_kernel(const unsigned short* __restrict__ in_planes, long in_stride, int ld_in,
                                                              const float* __restrict__ in_inv, const float* __restrict__ in_bias,
                                                              const unsigned short* __restrict__ out_planes, long out_stride,
                                                              int ld_out, const float* __restrict__ out_inv,
                                                              const float* __restrict__ out_bias, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int inter,
                                                              unsigned short* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)IMAGE_BYTES / 2;
    if (idx >= total) return;
    if (idx < VEC_BYTES / 2) {
        const int fi = (int)(idx >> 1);
        const float* vec = fi < 256 ? out_inv : fi < 512 ? out_bias : fi < 768 ? gamma : beta;
        const unsigned bits = __builtin_bit_cast(unsigned, vec[fi & 255]);
        img[idx] = (unsigned short)((idx & 1) ? (bits >> 16) : (bits & 0xffffu));
        return;
    }
    const long k = idx - VEC_BYTES / 2;
    const int e = (int)(k % 512), f = (int)((k / 512) % CHUNK_FRAGS), st = (int)(k / (512L * CHUNK_FRAGS));
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    unsigned short v16 = 0;
    if (st < 3 * NH) {
        int row0;
        if (inter) row0 = (st % 3) * D + (st / 3) * 32;
        else row0 = st < NH ? 2 * D + st * 32 : ((st - NH) & 1) * D + ((st - NH) >> 1) * 32;
        if (f < W_FRAGS) {
            const int s = f >> 2, hh = (f >> 1) & 1, pl = f & 1;
            v16 = in_planes[pl * in_stride + (size_t)(row0 + 16 * hh + m) * ld_in + 32 * s + 8 * kg + j];
        } else {
            // (q stages: x 1 / sqrt(32) x log2(e) -- the scores leave the MFMA ready for exp2)
            const bool is_q = inter ? (st % 3) == 0 : (st >= NH && ((st - NH) & 1) == 0);
            const float c = is_q ? 0.17677669529663688f * 1.4426950408889634f : 1.f;
            const int fi = e >> 1;
            float v = 0.f;
            if (fi < 32) v = in_inv[row0 + fi] * c;
            else if (fi < 64) v = in_bias ? in_bias[row0 + fi - 32] * c : 0.f;
            const unsigned bits = __builtin_bit_cast(unsigned, v);
            v16 = (unsigned short)((e & 1) ? (bits >> 16) : (bits & 0xffffu));
        }
    } else {
        const int hd = st - 3 * NH;
        if (f < W_FRAGS) {
            const int t = f >> 1, pl = f & 1;
            const int orow = 32 * (t >> 1) + 8 * (m >> 2) + 4 * (t & 1) + (m & 3);
            v16 = out_planes[pl * out_stride + (size_t)orow * ld_out + 32 * hd + 16 * (j >> 2) + 4 * kg + (j & 3)];
        }
    }
    img[idx] = v16;
}

// RAW stages behind a block image: stage c = rows 32 c .. 32 c + 31 of the [384, 256] offsets | logits weight; fragment 4 s + 2 Hh + p =
// plane p of Ws[32 c + 16 Hh + m][32 s + 8 kg + j] (the block's own output reaches this product in fragment order); fragment 32 =
// 1 / row scale | bias.
__global__ __launch_bounds__(256) void dec_attn2_raw_image_kernel(const unsigned short* __restrict__ planes, long stride, int ld,
                                                                  const float* __restrict__ inv, const float* __restrict__ bias,
                                                                  unsigned short* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)RAW_STAGES * CHUNK_FRAGS * 512;
    if (idx >= total) return;
    const int e = (int)(idx % 512), f = (int)((idx / 512) % CHUNK_FRAGS), c = (int)(idx / (512L * CHUNK_FRAGS));
    const int l = e >> 3, j = e & 7, m = l & 15, kg = l >> 4;
    unsigned short v16 = 0;
    if (f < W_FRAGS) {
        const int s = f >> 2, hh = (f >> 1) & 1, pl = f & 1;
        v16 = planes[pl * stride + (size_t)(32 * c + 16 * hh + m) * ld + 32 * s + 8 * kg + j];
    } else {
        const int fi = e >> 1;
        float v = 0.f;
        if (fi < 32) v = inv[32 * c + fi];
        else if (fi < 64) v = bias ? bias[32 * c + fi - 32] : 0.f;
        const unsigned bits = __builtin_bit_cast(unsigned, v);
        v16 = (unsigned short)((e & 1) ? (bits >> 16) : (bits & 0xffffu));
    }
    img[idx] = v16;
}

}  // namespace

#ifdef A2_STAMPS
extern "C" int gom_dec_attn2_set_stamps(void* device_buffer) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_a2_stamps), &device_buffer, sizeof(void*));
}
#endif

extern "C" long gom_dec_attn2_image_bytes(int d_model, int heads) {
    if (d_model != D || heads != NH) return -1;
    return IMAGE_BYTES;
}

extern "C" long gom_dec_attn2_raw_image_bytes(void) { return RAW_IMAGE_BYTES; }

extern "C" int gom_dec_attn2_image(const void* in_planes, long in_plane_stride, int ld_in, const float* in_inv_scale,
                                   const float* in_bias, const void* out_planes, long out_plane_stride, int ld_out,
                                   const float* out_inv_scale, const float* out_bias, const float* gamma, const float* beta,
                                   int inter, void* image, long image_bytes, void* stream) {
    GOM_CHECK_ARG(in_planes && in_inv_scale && out_planes && image && ld_in >= D && ld_out >= D);
    GOM_CHECK_ARG(out_inv_scale && out_bias && gamma && beta);
    GOM_CHECK_ARG(image_bytes >= IMAGE_BYTES);
    const long total = (long)IMAGE_BYTES / 2;
    hipLaunchKernelGGL(dec_attn2_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)in_planes, in_plane_stride, ld_in, in_inv_scale, in_bias,
                       (const unsigned short*)out_planes, out_plane_stride, ld_out, out_inv_scale, out_bias, gamma, beta,
                       inter ? 1 : 0, (unsigned short*)image);
    return gom_launch_status();
}

/* appends the offsets | logits stages to a block image of gom_dec_attn2_raw_image_bytes() whose head gom_dec_attn2_image has filled */
extern "C" int gom_dec_attn2_raw_image(const void* raw_planes, long raw_plane_stride, int ld_raw, const float* raw_inv_scale,
                                       const float* raw_bias, void* image, long image_bytes, void* stream) {
    GOM_CHECK_ARG(raw_planes && raw_inv_scale && image && ld_raw >= D && image_bytes >= RAW_IMAGE_BYTES);
    const long total = (long)RAW_STAGES * CHUNK_FRAGS * 512;
    hipLaunchKernelGGL(dec_attn2_raw_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)raw_planes, raw_plane_stride, ld_raw, raw_inv_scale, raw_bias,
                       (unsigned short*)((unsigned char*)image + IMAGE_BYTES));
    return gom_launch_status();
}

extern "C" int gom_dec_attn2_f32(const float* X, int ldx, const float* P, int ldp, const void* image, float eps, float* Y, int ldy,
                                 int groups, int group_tokens, int inner, int inter, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && Y && groups >= 0 && group_tokens > 0);
    GOM_CHECK_ARG(ldx >= D && ldy >= D && (ldx % 4) == 0 && (ldy % 4) == 0 && (!P || (ldp >= D && (ldp % 4) == 0)));
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && (!P || ((uintptr_t)P % 16) == 0) && ((uintptr_t)Y % 16) == 0 &&
                  ((uintptr_t)image % 16) == 0);
    GOM_CHECK_ARG(inter ? (!P && inner > 0 && group_tokens <= 128) : (P && group_tokens <= 32));
    if (groups == 0) return GOM_OK;
    DecArgs2 a{};
    a.X = X; a.P = P; a.img = (const unsigned char*)image; a.Y = Y; a.flag = flag; a.eps = eps; a.scale = 1.0f / sqrtf(32.f);
    a.ldx = ldx; a.ldp = ldp; a.ldy = ldy; a.groups = groups; a.G = group_tokens; a.per_wave = cdiv(group_tokens, WAVES); a.inner = inner;
    hipError_t e = hipFuncSetAttribute((const void*)dec_attn2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dec_attn2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    if (inter) hipLaunchKernelGGL(dec_attn2_kernel<true>, dim3((unsigned)groups), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dec_attn2_kernel<false>, dim3((unsigned)cdiv(groups, 4)), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return gom_launch_status();
}

extern "C" int gom_dec_attn2_raw_f32(const float* X, int ldx, const void* image, float eps, float* Y, int ldy, const float* P2, int ldp2,
                                     float* raw, int ldraw, int groups, int group_tokens, int inner, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && Y && P2 && raw && groups >= 0 && group_tokens > 0 && group_tokens <= 128 && inner > 0);
    GOM_CHECK_ARG(ldx >= D && ldy >= D && ldp2 >= D && ldraw >= 384 && (ldx % 4) == 0 && (ldy % 4) == 0 && (ldp2 % 4) == 0 && (ldraw % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)P2 % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)raw % 16) == 0 &&
                  ((uintptr_t)image % 16) == 0);
    if (groups == 0) return GOM_OK;
    DecArgs2 a{};
    a.X = X; a.P = nullptr; a.img = (const unsigned char*)image; a.Y = Y; a.flag = flag; a.eps = eps; a.scale = 1.0f / sqrtf(32.f);
    a.ldx = ldx; a.ldp = 0; a.ldy = ldy; a.groups = groups; a.G = group_tokens; a.per_wave = cdiv(group_tokens, WAVES); a.inner = inner;
    a.P2 = P2; a.ldp2 = ldp2; a.RAWO = raw; a.ldraw = ldraw;
    hipError_t e = hipFuncSetAttribute((const void*)dec_attn2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    hipLaunchKernelGGL((dec_attn2_kernel<true, true>), dim3((unsigned)groups), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return gom_launch_status();
}
