// The two self-attention blocks of a DeepSolo composite decoder layer, each as ONE launch (f16x3 split, fp32-class accuracy):
//
//   intra:  tgt = norm_intra(tgt + out_proj(MHA(q = k = tgt + query_pos, v = tgt)))   over the 25 points of one query
//   inter:  tgt = norm_inter(tgt + out_proj(MHA(q = k = v = tgt)))                    over the nq queries of one (frame, point)
//
// (/root/reference/third_party/adet/layers/deformable_transformer.py:386-404; nn.MultiheadAttention, 8 heads of 32, eval mode).
// Both are LOCAL problems -- 25 x 256 resp. nq x 256 tokens -- that used to run as five launches through HBM each (q|k
// projection, v projection, attention core, out_proj + LayerNorm; 140 + 200 us per layer at 8 x 100 x 25 rows, the Q-side
// products at 0.15 of the f16x3 MFMA peak).  Here a token never leaves its lane between the first load and the final store:
//
//   * a wave owns up to 32 tokens of ONE attention group (intra: the 25 points of a query; inter: a quarter of the queries of
//     a (frame, point)) as MFMA operand fragments in 128 VGPRs, as in gemm_k256.hip / ffn_fused.hip; a workgroup = 4 waves =
//     four intra groups or one inter group; one wave per SIMD with the whole register file;
//   * the in-projection weights stream through a three-slot LDS ring by MUBUF LDS-DMA from a fragment-linear image (stage i + 2
//     is requested while stage i computes; nine fragments per wave and stage, so the end-of-stage wait is a counted one), one
//     36 KB stage per (head, q | k | v): q and k are computed TRANSPOSED (lane = token, registers = the head's 32 features), v STRAIGHT
//     (lane = feature, registers = tokens) -- which are exactly the operand layouts the attention products want:
//         S^T[key, query] = K . Q^T     A = K accumulators (lane = key), B = Q accumulators (lane = query); the k index runs
//                                       over the head's features in ACCUMULATOR order, the same permutation on both sides
//         O^T[d,  query]  = V^T . P^T   A = V accumulators (lane = d, registers = keys), B = P straight from S^T's registers
//     after an fp16 two-plane split of the registers -- no shuffle, no LDS round trip (the fused FFN kernel's trick); softmax
//     statistics are per-lane scalars + one exchange between the half-waves;
//   * inter: the four waves exchange the K and V^T fragments of a head through 32 KB of LDS (written once, read by all);
//   * O^T (lane = token) is the B operand of out_proj, whose weights follow in eight k-major stages whose k order is the
//     accumulator order of O^T (baked into the image); Y^T's statistics are in-lane sums, so residual + LayerNorm finish in
//     registers and every lane stores its token's 16-byte pieces.
// Plane products in the tile kernel's order (x-lo w-hi, x-hi w-lo, x-hi w-hi per 16-wide k-step); activations must stay within
// fp16's range (|x| <= 65504: q, k, v and the inputs are checked, *flag is raised otherwise -- gemm_f16x3.hip's contract).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 256, NH = 8;                           // model width, heads (head dim 32 = one 32-column chunk)
constexpr int FRAG = 1024;                               // bytes of one MFMA operand fragment
constexpr int W_FRAGS = 32;                              // weight fragments of a stage
constexpr int CHUNK_FRAGS = W_FRAGS + 4;                 // + (1 / row scale | bias) of the chunk's 32 columns + 3 unused: 9 per wave
constexpr int CHUNK_BYTES = CHUNK_FRAGS * FRAG;          // 36 KB
constexpr int STAGES = 3 * NH + NH;                      // 24 projection chunks + 8 out_proj stages
constexpr int SLOTS = 3;                                 // ring depth: stage i + 2 is requested while stage i computes
constexpr int RING_BYTES = SLOTS * CHUNK_BYTES;
constexpr int XCH_BYTES = 4 * 8 * FRAG;                  // inter: K (4) + V^T (4) fragments of each of the four waves, one head
constexpr int IMAGE_BYTES = STAGES * CHUNK_BYTES;
constexpr int RAW_STAGES = 12;                           // RAW form: the cross attention's offsets | logits product (N = 384) behind the block
constexpr int RAW_IMAGE_BYTES = (STAGES + RAW_STAGES) * CHUNK_BYTES;

struct DecArgs {
    const float* X;                                          // tgt [rows, 256]
    const float* P;                                          // query_pos (intra) or null
    const unsigned char* img;
    float* Y;
    int* flag;
    float eps, scale;
    int ldx, ldp, ldy;
    int groups;                                              // attention groups of the launch
    int G;                                                   // tokens per group
    int per_wave;                                            // inter: tokens of a group per wave = ceil(G / 4)
    int inner;                                               // inter: rows between consecutive tokens of a group (= points)
    // RAW form: raw = (Y + P2) Wraw^T + braw, the sampling offsets | attention logits of the cross attention that follows
    const float* P2;                                         // query_pos [rows, 256]
    float* RAWO;                                             // [rows, 384]
    int ldp2, ldraw;
};

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

// `lane_off` (vector) = this lane's 16 bytes inside a fragment, `frag_off` (scalar) = the fragment's offset in the image
__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned frag_off, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)lane_off, (int)frag_off, 0, 0);
}

// registers 8 s .. 8 s + 7 of a 32x32 accumulator -> the two planes of k-step s of an operand fragment
__device__ __forceinline__ void acc_to_frags(const f32x16& a, half8 (&f)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        split8(f32x4{a[8 * s], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3]},
               f32x4{a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7]}, f[s][0], f[s][1]);
}

// C (+)= A . B on the f16x3 planes: a / b = [k-step][plane hi, lo]
__device__ __forceinline__ f32x16 mfma_x3(const half8 a_hi, const half8 a_lo, const half8 b_hi, const half8 b_lo, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, c, 0, 0, 0);
    return c;
}

// one weight stage: 32 fragments in four groups of eight, group g + 1 read while the MFMAs of group g issue (two-deep register
// pipeline pinned with sched_group_barrier); the next stage's fragments of this wave are requested one per four MFMAs
#define DA_LOAD(dst, g)                                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                          \
        dst[i_] = *reinterpret_cast<const half8*>(base + ((g) * 8 + i_) * FRAG);
#define DA_DMA(i) dma_fragment(rs_img, lane16, nsrc + (i) * 4 * FRAG, ndst + (i) * 4 * FRAG);
#define DA_PIN3()                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#define DA_PIN2()                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#define DA_STAGE(MFMA)                                                                                        \
    {                                                                                                         \
        half8 fa[8], fb[8];                                                                                   \
        DA_LOAD(fa, 0)                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                    \
        DA_LOAD(fb, 1) MFMA(fa, 0) DA_DMA(0) DA_DMA(1) DA_DMA(2) DA_PIN3()                                    \
        DA_LOAD(fa, 2) MFMA(fb, 1) DA_DMA(3) DA_DMA(4) DA_DMA(5) DA_PIN3()                                    \
        DA_LOAD(fb, 3) MFMA(fa, 2) DA_DMA(6) DA_DMA(7) DA_DMA(8) DA_PIN3()                                    \
        MFMA(fb, 3)                                                                                           \
    }

#define DA_STAGE_LAST(MFMA)                                                                                   \
    {                                                                                                         \
        half8 fa[8], fb[8];                                                                                   \
        DA_LOAD(fa, 0)                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                    \
        DA_LOAD(fb, 1) MFMA(fa, 0) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); \
        DA_LOAD(fa, 2) MFMA(fb, 1) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); \
        DA_LOAD(fb, 3) MFMA(fa, 2) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); \
        MFMA(fb, 3)                                                                                           \
    }

// projection chunk, transposed: acc[feature][token] += W chunk . X^T  (A = weight fragment, B = the rows in registers)
#define DA_MFMA_T(src, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int s_ = (g) * 4 + i_;                                                                          \
        acc = mfma_x3(src[2 * i_], src[2 * i_ + 1], xf[0][s_], xf[1][s_], acc);                               \
    }
// projection chunk, straight: acc[token][feature] += X . W chunk^T  (A = the rows in registers, B = weight fragment)
#define DA_MFMA_S(src, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int s_ = (g) * 4 + i_;                                                                          \
        acc = mfma_x3(xf[0][s_], xf[1][s_], src[2 * i_], src[2 * i_ + 1], acc);                               \
    }
// out_proj stage of head `hh_`: fragments [tile][k-step][plane]; group g = output tiles 2 g, 2 g + 1
#define DA_MFMA_O(src, g)                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
        const int t_ = 2 * (g) + (i_ >> 1), s_ = i_ & 1;                                                      \
        yacc[t_] = mfma_x3(src[2 * i_], src[2 * i_ + 1], of[hh_][s_][0], of[hh_][s_][1], yacc[t_]);           \
    }

template <bool INTER, bool RAW = false>
__global__ __launch_bounds__(256, 1) void dec_attn_kernel(const DecArgs p) {
    constexpr int NST = STAGES + (RAW ? RAW_STAGES : 0);     // stages of the image this instantiation streams
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xch = smem + RING_BYTES;                  // inter only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    constexpr int NB = INTER ? 4 : 1;                        // key blocks (waves) of an attention group

    // ---- which token this lane owns ----
    long row;
    bool valid;
    int ntok[NB];                                            // valid key slots per block
    if constexpr (!INTER) {
        const long gi = (long)blockIdx.x * 4 + wave;
        const long g = gi < p.groups ? gi : p.groups - 1;
        ntok[0] = p.G;
        valid = gi < p.groups && fr < p.G;
        row = g * p.G + (fr < p.G ? fr : 0);
    } else {
        const long gi = blockIdx.x;
        const long b = gi / p.inner, pp = gi % p.inner;
#pragma unroll
        for (int w = 0; w < NB; ++w) {
            const int left = p.G - w * p.per_wave;
            ntok[w] = left < 0 ? 0 : (left < p.per_wave ? left : p.per_wave);
        }
        const int mine = p.G - wave * p.per_wave;
        valid = fr < p.per_wave && fr < mine;
        const long tq = valid ? (long)wave * p.per_wave + fr : 0;
        row = (b * p.G + tq) * p.inner + pp;
    }

    unsigned pf[2];
    gom_prefetch_image(p.img, (unsigned)(NST * CHUNK_BYTES), tid, 256, pf);                 // (common.h: a one-round launch, the image cold)
    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, NST * CHUNK_BYTES, 0x00020000);
    constexpr unsigned OOB = 0x7FFF0000u;                    // beyond num_records: the DMA writes zeros (into an unused stage)
    const unsigned lane16 = lane * 16;

    // range bookkeeping: a running maximum of |value| over everything that is split into fp16 planes, and a NaN / Inf detector
    // over the outputs (o * 0 accumulates to NaN for a non-finite o).  NOT `bad |= !(|v| <= limit)` per value: the compiler defers
    // those compares to the end of the kernel and keeps (spills) every compared value until then -- 1000 registers of scratch.
    float amax = 0.f, chk = 0.f;
    half8 xf[2][D / 16];
    // this wave's rows as operand fragments: lane (r, h) holds x[row r][16 s + 8 h .. + 7], two planes -- whole-line loads and a
    // layout change in a ring slot that holds no stage at that moment (common.h gom_rows_to_fragments).  Token slot r of the wave
    // (0..31) -> its row; slots beyond the group's tokens recompute token 0 of the group (masked as keys, never stored)
    auto slot_row = [&](int r) -> long {
        if constexpr (!INTER) {
            const long gi = (long)blockIdx.x * 4 + wave;
            const long g = gi < p.groups ? gi : p.groups - 1;
            return g * p.G + (r < p.G ? r : 0);
        } else {
            const long gi = blockIdx.x;
            const long b = gi / p.inner, pp = gi % p.inner;
            const int mine = p.G - wave * p.per_wave;
            const long tq = (r < p.per_wave && r < mine) ? (long)wave * p.per_wave + r : 0;
            return (b * p.G + tq) * p.inner + pp;
        }
    };
    auto xrow = [&](int r) { return p.X + (size_t)slot_row(r) * p.ldx; };
    auto prow = [&](int r) { return p.P + (size_t)slot_row(r) * p.ldp; };
    {
        // kernel start: the ring's first two stages are requested behind the first loads, slot 2 is the scratch
        float* scratch = reinterpret_cast<float*>(smem + 2 * CHUNK_BYTES) + wave * (32 * 64);
        gom_rows_to_fragments<64, false>(xrow, xrow, scratch, lane, xf, amax, [&]() {
            for (int f = wave; f < 2 * CHUNK_FRAGS; f += 4) dma_fragment(rs_img, lane16, f * FRAG, smem + f * FRAG);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gom_prefetch_done(pf);
    __syncthreads();

    // per stage: `base` = this lane's slice of the stage in the ring, `nsrc` / `ndst` = this wave's fragments of the next one
#define DA_STAGE_VARS(i)                                                                                      \
    const unsigned char* base = smem + ((i) % SLOTS) * CHUNK_BYTES + lane * 16;                               \
    const float* aux = reinterpret_cast<const float*>(smem + ((i) % SLOTS) * CHUNK_BYTES + W_FRAGS * FRAG);   \
    const unsigned nsrc = (i) + 2 < NST ? (unsigned)((i) + 2) * CHUNK_BYTES + wave * FRAG : OOB;             \
    unsigned char* ndst = smem + (((i) + 2) % SLOTS) * CHUNK_BYTES + wave * FRAG;
    // end of a stage: everything older than this stage's nine requests (= stage i + 1, requested a stage ago) has landed --
    // loads return in issue order; stages that issue other vector-memory operations behind their requests wait for all
#define DA_STAGE_END()                                                                                        \
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                                                          \
    __syncthreads();                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define DA_STAGE_END_ALL()                                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
    __syncthreads();                                                                                          \
    __builtin_amdgcn_sched_barrier(0);

    // transposed chunk epilogue: value = acc * (1 / row scale) + bias, features (g & 3) + 8 (g >> 2) + 4 fh of the chunk
    auto finish_t = [&](f32x16& acc, const float* aux) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + 8 * q + 4 * fh);
            const f32x4 bi = *reinterpret_cast<const f32x4*>(aux + 32 + 8 * q + 4 * fh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[4 * q + e] = fmaf(acc[4 * q + e], sc[e], bi[e]);
                amax = fmaxf(amax, fabsf(acc[4 * q + e]));
            }
        }
        asm volatile("" : "+v"(amax));
    };
    // straight chunk epilogue: the lane is the feature
    auto finish_s = [&](f32x16& acc, const float* aux) {
        const float sc = aux[fr], bi = aux[32 + fr];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            acc[g] = fmaf(acc[g], sc, bi);
            amax = fmaxf(amax, fabsf(acc[g]));
        }
        asm volatile("" : "+v"(amax));
    };

    half8 of[NH][2][2];                                      // O^T of every head as out_proj's B operand: [head][k-step][plane]

    // softmax(scale * S^T) over the keys of the group (registers of `s` <-> key slot (g & 3) + 8 (g >> 2) + 4 fh of block b),
    // unnormalised probabilities left in `s`, returns 1 / sum
    auto softmax_keys = [&](f32x16 (&s)[NB]) {
        float mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = (g & 3) + 8 * (g >> 2) + 4 * fh;
                s[b][g] = j < ntok[b] ? s[b][g] * p.scale : -INFINITY;
                mx = fmaxf(mx, s[b][g]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                s[b][g] = __expf(s[b][g] - mx);               // v_exp_f32: ~1 ulp, far inside the block's fp32-class error
                sum += s[b][g];
            }
        sum += __shfl_xor(sum, 32, 64);
        return 1.f / sum;
    };

    if constexpr (!INTER) {
        // ---- sweep 1: V of every head (stages 0 .. 7), kept as A-operand fragments in of[h] until the head's attention ----
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            DA_STAGE_VARS(h)
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.f;
            DA_STAGE(DA_MFMA_S)
            finish_s(acc, aux);
            acc_to_frags(acc, of[h]);
            DA_STAGE_END()
        }
        // ---- sweep 2: q | k of (tgt + query_pos) per head (stages 8 + 2 h, 9 + 2 h), attention in registers ----
        __builtin_amdgcn_sched_barrier(0);
        {
            // (stage 8 sits in slot 2, stage 9 in slot 0; slot 1 held stage 7 and is free until stage 8 requests stage 10 into it)
            float* scratch = reinterpret_cast<float*>(smem + 1 * CHUNK_BYTES) + wave * (32 * 64);
            gom_rows_to_fragments<64, true>(xrow, prow, scratch, lane, xf, amax, [&]() {});
        }
        __syncthreads();                                     // stage 8's product requests stage 10 into slot 1: every wave's scratch
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            half8 qf[2][2], kf[2][2];
            {
                DA_STAGE_VARS(NH + 2 * h)
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
                DA_STAGE(DA_MFMA_T)
                finish_t(acc, aux);
                acc_to_frags(acc, qf);
                DA_STAGE_END()
            }
            {
                DA_STAGE_VARS(NH + 2 * h + 1)
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
                DA_STAGE(DA_MFMA_T)
                finish_t(acc, aux);
                acc_to_frags(acc, kf);
                DA_STAGE_END()
            }
            f32x16 s[1];
#pragma unroll
            for (int g = 0; g < 16; ++g) s[0][g] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[0] = mfma_x3(kf[ks][0], kf[ks][1], qf[ks][0], qf[ks][1], s[0]);
            const float inv = softmax_keys(s);
            half8 pf[2][2];
            acc_to_frags(s[0], pf);
            f32x16 o;
#pragma unroll
            for (int g = 0; g < 16; ++g) o[g] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) o = mfma_x3(of[h][ks][0], of[h][ks][1], pf[ks][0], pf[ks][1], o);
#pragma unroll
            for (int g = 0; g < 16; ++g) o[g] *= inv;
            acc_to_frags(o, of[h]);                          // V of this head is dead: its registers take O^T
        }
    } else {
        // ---- per head: q, k, v of tgt (stages 3 h, 3 h + 1, 3 h + 2); K and V^T fragments shared through LDS ----
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            half8 qf[2][2];
            {
                DA_STAGE_VARS(3 * h)
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
                DA_STAGE(DA_MFMA_T)
                finish_t(acc, aux);
                acc_to_frags(acc, qf);
                DA_STAGE_END()
            }
            {
                DA_STAGE_VARS(3 * h + 1)
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
                DA_STAGE(DA_MFMA_T)
                finish_t(acc, aux);
                half8 kf[2][2];
                acc_to_frags(acc, kf);
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    *reinterpret_cast<half8*>(xch + (wave * 8 + f) * FRAG + lane * 16) = kf[f >> 1][f & 1];
                DA_STAGE_END()
            }
            {
                DA_STAGE_VARS(3 * h + 2)
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.f;
                DA_STAGE(DA_MFMA_S)
                finish_s(acc, aux);
                half8 vf[2][2];
                acc_to_frags(acc, vf);
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    *reinterpret_cast<half8*>(xch + (wave * 8 + 4 + f) * FRAG + lane * 16) = vf[f >> 1][f & 1];
                DA_STAGE_END()
            }
            f32x16 s[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int g = 0; g < 16; ++g) s[b][g] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const half8 k_hi = *reinterpret_cast<const half8*>(xch + (b * 8 + 2 * ks) * FRAG + lane * 16);
                    const half8 k_lo = *reinterpret_cast<const half8*>(xch + (b * 8 + 2 * ks + 1) * FRAG + lane * 16);
                    s[b] = mfma_x3(k_hi, k_lo, qf[ks][0], qf[ks][1], s[b]);
                }
            }
            const float inv = softmax_keys(s);
            f32x16 o;
#pragma unroll
            for (int g = 0; g < 16; ++g) o[g] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                half8 pf[2][2];
                acc_to_frags(s[b], pf);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const half8 v_hi = *reinterpret_cast<const half8*>(xch + (b * 8 + 4 + 2 * ks) * FRAG + lane * 16);
                    const half8 v_lo = *reinterpret_cast<const half8*>(xch + (b * 8 + 4 + 2 * ks + 1) * FRAG + lane * 16);
                    o = mfma_x3(v_hi, v_lo, pf[ks][0], pf[ks][1], o);
                }
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) o[g] *= inv;
            acc_to_frags(o, of[h]);
        }
    }

    // ---- out_proj: Y^T[256 x tokens] += Wo[:, head h's features] . O_h^T, eight stages ----
    f32x16 yacc[D / 32];
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) yacc[t][g] = 0.f;
    // the residual rows (= X, last read at the kernel's start) are requested in front of the first out_proj stage, into the
    // registers the row fragments no longer need; epilogue vectors (1 / row scale, bias, gamma, beta) ride in the unused
    // fragments of the last two stages
    f32x4 res[D / 8];
    {
        const float* rr = p.X + (size_t)row * p.ldx + 4 * fh;
#pragma unroll
        for (int i = 0; i < D / 8; ++i) res[i] = *reinterpret_cast<const f32x4*>(rr + 8 * i);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int hh_ = h;
        DA_STAGE_VARS(3 * NH + h)
        (void)aux;
        if (RAW || h < NH - 2) {                             // (RAW: twelve more stages follow, the ring keeps running)
            DA_STAGE(DA_MFMA_O)
            DA_STAGE_END()
        } else {
            (void)nsrc;
            (void)ndst;
            DA_STAGE_LAST(DA_MFMA_O)
            if (h == NH - 2) { DA_STAGE_END_ALL() }
        }
    }

    // ---- residual + LayerNorm in registers: lane (token, fh) holds features 32 t + 8 q + 4 fh .. + 3 of its token ----
    {
        const float* v_inv = reinterpret_cast<const float*>(smem + ((STAGES - 1) % SLOTS) * CHUNK_BYTES + (W_FRAGS + 1) * FRAG);
        const float* v_bias = v_inv + 256;
        const float* v_gamma = v_inv + 512;
        // (RAW: stage 30's slot is already being refilled with stage 33 -- beta rides in stage 31 as well, fragment 32)
        const float* v_beta = RAW ? reinterpret_cast<const float*>(smem + ((STAGES - 1) % SLOTS) * CHUNK_BYTES + W_FRAGS * FRAG)
                                  : reinterpret_cast<const float*>(smem + ((STAGES - 2) % SLOTS) * CHUNK_BYTES + (W_FRAGS + 1) * FRAG);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * t + 8 * q + 4 * fh;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(v_inv + col);
                const f32x4 bi = *reinterpret_cast<const f32x4*>(v_bias + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(yacc[t][4 * q + e], sc[e], bi[e]) + res[4 * t + q][e];
                    yacc[t][4 * q + e] = v;
                    sum += v;
                }
            }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.f / D);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                yacc[t][g] -= mean;
                sq += yacc[t][g] * yacc[t][g];
            }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.f / D) + p.eps);
        float* yr = p.Y + (size_t)row * p.ldy + 4 * fh;
        const float* pr = RAW ? p.P2 + (size_t)row * p.ldp2 + 4 * fh : nullptr;
#pragma unroll
        for (int t = 0; t < D / 32; ++t) {
            f32x16 x2;                                       // RAW: (block output + query_pos) of tile t, accumulator order
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * t + 8 * q + 4 * fh;
                const f32x4 ga = *reinterpret_cast<const f32x4*>(v_gamma + col);
                const f32x4 be = *reinterpret_cast<const f32x4*>(v_beta + col);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = yacc[t][4 * q + e] * rstd * ga[e] + be[e];
                    chk = fmaf(o[e], 0.f, chk);
                }
                if (valid) *reinterpret_cast<f32x4*>(yr + 32 * t + 8 * q) = o;
                if constexpr (RAW) {
                    const f32x4 pq = *reinterpret_cast<const f32x4*>(pr + 32 * t + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x2[4 * q + e] = o[e] + pq[e];
                        amax = fmaxf(amax, fabsf(x2[4 * q + e]));
                    }
                }
            }
            if constexpr (RAW) {
                // the rows of the NEXT product as B-operand fragments: registers 8 s .. 8 s + 7 of tile t = k-step 2 t + s in accumulator
                // order (the order out_proj's image uses for O^T; baked into the raw stages' image)
                half8 f2[2][2];
                acc_to_frags(x2, f2);
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    xf[0][2 * t + s_] = f2[s_][0];
                    xf[1][2 * t + s_] = f2[s_][1];
                }
            }
        }
        asm volatile("" : "+v"(chk), "+v"(amax));
    }
    if constexpr (RAW) {
        // ---- raw = (Y + query_pos) Wraw^T + braw: twelve 32-column chunks, transposed (lane = token), stored as 16-byte pieces ----
        __builtin_amdgcn_sched_barrier(0);
        // A chunk's four stores are issued at the START of the next chunk (its values wait in 16 registers): they are then the
        // oldest vector-memory operations of that stage, and its counted end-of-stage wait never waits for a store just issued
        // (with the stores behind the stage's requests and a wait for all, a stage took 2.3 us instead of ~1).
        float* rr = p.RAWO + (size_t)row * p.ldraw + 4 * fh;
        f32x4 pend[4];
#pragma unroll
        for (int c = 0; c < RAW_STAGES; ++c) {
            DA_STAGE_VARS(STAGES + c)
            if (c > 0 && valid) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(rr + 32 * (c - 1) + 8 * q) = pend[q];
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.f;
            if (c < RAW_STAGES - 2) {
                DA_STAGE(DA_MFMA_T)
            } else {
                (void)nsrc;
                (void)ndst;
                DA_STAGE_LAST(DA_MFMA_T)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(aux + 8 * q + 4 * fh);
                const f32x4 bi = *reinterpret_cast<const f32x4*>(aux + 32 + 8 * q + 4 * fh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pend[q][e] = fmaf(acc[4 * q + e], sc[e], bi[e]);
                    chk = fmaf(pend[q][e], 0.f, chk);
                }
            }
            asm volatile("" : "+v"(chk));
            if (c < RAW_STAGES - 2) { DA_STAGE_END() }
            else if (c < RAW_STAGES - 1) { DA_STAGE_END_ALL() }
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(rr + 32 * (RAW_STAGES - 1) + 8 * q) = pend[q];
        }
    }
    // an operand left fp16's range, or a result is not finite (gemm_f16x3.hip contract; fmaxf drops a NaN, `chk` catches it)
    if ((!(amax <= 65504.f) || !(chk == 0.f)) && p.flag) atomicOr(p.flag, 1);
}

#undef DA_LOAD
#undef DA_DMA
#undef DA_PIN3
#undef DA_PIN2
#undef DA_STAGE
#undef DA_MFMA_T
#undef DA_MFMA_S
#undef DA_MFMA_O
#undef DA_STAGE_VARS
#undef DA_STAGE_END

// Fragment-linear image of a block's weights (row-scaled planes of gom_split_f16x2): in_proj W [768, 256] = q | k | v rows,
// out_proj W [256, 256].  Stage order -- intra: v_0 .. v_7, then (q_h, k_h) for h = 0 .. 7; inter: (q_h, k_h, v_h) for
// h = 0 .. 7; then the eight out_proj stages.  A projection stage = gemm_k256.hip's chunk of the 32 weight rows of (head, q | k |
// v): fragment f = 2 s + p: element j of lane (r, h) = plane p of Ws[row0 + r][16 s + 8 h + j]; fragment 32: floats 0..31 =
// 1 / row scale, 32..63 = bias.  out_proj stage hd: fragment (t * 2 + s) * 2 + p: element j of lane (r, h) = plane p of
// Wo[32 t + r][32 hd + (j & 3) + 8 (2 s + (j >> 2)) + 4 h] -- O^T's accumulator order of head hd's features.
__global__ __launch_bounds__(256) void dec_attn_image_kernel(const unsigned short* __restrict__ in_planes, long in_stride, int ld_in,
                                                             const float* __restrict__ in_inv, const float* __restrict__ in_bias,
                                                             const unsigned short* __restrict__ out_planes, long out_stride,
                                                             int ld_out, const float* __restrict__ out_inv,
                                                             const float* __restrict__ out_bias, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int inter,
                                                             unsigned short* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)STAGES * CHUNK_FRAGS * 512;
    if (idx >= total) return;
    const int e = (int)(idx % 512), f = (int)((idx / 512) % CHUNK_FRAGS), st = (int)(idx / (512L * CHUNK_FRAGS));
    const int l = e >> 3, j = e & 7, r = l & 31, h = l >> 5;
    if (st < 3 * NH) {
        int row0;
        if (inter) row0 = (st % 3) * D + (st / 3) * 32;      // q | k | v of head st / 3
        else row0 = st < NH ? 2 * D + st * 32 : ((st - NH) & 1) * D + ((st - NH) >> 1) * 32;
        if (f < W_FRAGS) {
            const int s = f >> 1, pl = f & 1;
            img[idx] = in_planes[pl * in_stride + (size_t)(row0 + r) * ld_in + 16 * s + 8 * h + j];
        } else if (f == W_FRAGS) {
            const int fi = e >> 1;
            float v = 0.f;
            if (fi < 32) v = in_inv[row0 + fi];
            else if (fi < 64) v = in_bias ? in_bias[row0 + fi - 32] : 0.f;
            const unsigned bits = __builtin_bit_cast(unsigned, v);
            img[idx] = (unsigned short)((e & 1) ? (bits >> 16) : (bits & 0xffffu));
        } else {
            img[idx] = 0;
        }
    } else {
        const int hd = st - 3 * NH;
        if (f < W_FRAGS) {
            const int pl = f & 1, s = (f >> 1) & 1, t = f >> 2;
            img[idx] = out_planes[pl * out_stride + (size_t)(32 * t + r) * ld_out + 32 * hd + (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * h];
        } else {
            // unused fragments of the last two stages: stage 31: 1 / row scale | bias | gamma (fragments 33, 34, 35), stage 30:
            // beta (fragment 33) -- 256 floats each, read by the epilogue straight from the ring
            const float* vec = nullptr;
            if (st == STAGES - 1) vec = f == W_FRAGS + 1 ? out_inv : f == W_FRAGS + 2 ? out_bias : f == W_FRAGS + 3 ? gamma : beta;   // (f == W_FRAGS: beta again, for the RAW form)
            if (st == STAGES - 2 && f == W_FRAGS + 1) vec = beta;
            unsigned short v16 = 0;
            if (vec) {
                const unsigned bits = __builtin_bit_cast(unsigned, vec[e >> 1]);
                v16 = (unsigned short)((e & 1) ? (bits >> 16) : (bits & 0xffffu));
            }
            img[idx] = v16;
        }
    }
}

// The RAW form's twelve extra stages behind a block's image: chunk c = rows 32 c .. 32 c + 31 of the [384, 256] offsets | logits
// weight; fragment f = 2 S + p (S = 0..15): element j of lane (r, h) = plane p of Ws[32 c + r][32 (S >> 1) + 16 (S & 1) + 8 (j >> 2) + 4 h
// + (j & 3)] -- the accumulator order in which the block's own output becomes this product's operand; fragment 32 = 1 / row scale | bias.
__global__ __launch_bounds__(256) void dec_attn_raw_image_kernel(const unsigned short* __restrict__ planes, long stride, int ld,
                                                                 const float* __restrict__ inv, const float* __restrict__ bias,
                                                                 unsigned short* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)RAW_STAGES * CHUNK_FRAGS * 512;
    if (idx >= total) return;
    const int e = (int)(idx % 512), f = (int)((idx / 512) % CHUNK_FRAGS), c = (int)(idx / (512L * CHUNK_FRAGS));
    const int l = e >> 3, j = e & 7, r = l & 31, h = l >> 5;
    unsigned short v16 = 0;
    if (f < W_FRAGS) {
        const int S = f >> 1, pl = f & 1;
        v16 = planes[pl * stride + (size_t)(32 * c + r) * ld + 32 * (S >> 1) + 16 * (S & 1) + 8 * (j >> 2) + 4 * h + (j & 3)];
    } else if (f == W_FRAGS) {
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

extern "C" long gom_dec_attn_raw_image_bytes(void) { return RAW_IMAGE_BYTES; }

/* appends the offsets | logits stages to a block image of RAW_IMAGE_BYTES whose first IMAGE_BYTES gom_dec_attn_image has filled */
extern "C" int gom_dec_attn_raw_image(const void* raw_planes, long raw_plane_stride, int ld_raw, const float* raw_inv_scale,
                                      const float* raw_bias, void* image, long image_bytes, void* stream) {
    GOM_CHECK_ARG(raw_planes && raw_inv_scale && image && ld_raw >= D && image_bytes >= RAW_IMAGE_BYTES);
    const long total = (long)RAW_STAGES * CHUNK_FRAGS * 512;
    hipLaunchKernelGGL(dec_attn_raw_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)raw_planes, raw_plane_stride, ld_raw, raw_inv_scale, raw_bias,
                       (unsigned short*)((unsigned char*)image + IMAGE_BYTES));
    return gom_launch_status();
}

extern "C" long gom_dec_attn_image_bytes(int d_model, int heads) {
    if (d_model != D || heads != NH) return -1;
    return IMAGE_BYTES;
}

extern "C" int gom_dec_attn_image(const void* in_planes, long in_plane_stride, int ld_in, const float* in_inv_scale,
                                  const float* in_bias, const void* out_planes, long out_plane_stride, int ld_out,
                                  const float* out_inv_scale, const float* out_bias, const float* gamma, const float* beta,
                                  int inter, void* image, long image_bytes, void* stream) {
    GOM_CHECK_ARG(in_planes && in_inv_scale && out_planes && image && ld_in >= D && ld_out >= D);
    GOM_CHECK_ARG(out_inv_scale && out_bias && gamma && beta);
    GOM_CHECK_ARG(image_bytes >= IMAGE_BYTES);
    const long total = (long)STAGES * CHUNK_FRAGS * 512;
    hipLaunchKernelGGL(dec_attn_image_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)in_planes, in_plane_stride, ld_in, in_inv_scale, in_bias,
                       (const unsigned short*)out_planes, out_plane_stride, ld_out, out_inv_scale, out_bias, gamma, beta,
                       inter ? 1 : 0, (unsigned short*)image);
    return gom_launch_status();
}

extern "C" int gom_dec_attn_f32(const float* X, int ldx, const float* P, int ldp, const void* image, float eps, float* Y, int ldy,
                                int groups, int group_tokens, int inner, int inter, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && Y && groups >= 0 && group_tokens > 0);
    GOM_CHECK_ARG(ldx >= D && ldy >= D && (ldx % 4) == 0 && (ldy % 4) == 0 && (!P || (ldp >= D && (ldp % 4) == 0)));
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && (!P || ((uintptr_t)P % 16) == 0) && ((uintptr_t)Y % 16) == 0 &&
                  ((uintptr_t)image % 16) == 0);
    GOM_CHECK_ARG(inter ? (!P && inner > 0 && group_tokens <= 128) : (P && group_tokens <= 32));
    if (groups == 0) return GOM_OK;
    DecArgs a{};
    a.X = X; a.P = P; a.img = (const unsigned char*)image; a.Y = Y; a.flag = flag; a.eps = eps; a.scale = 1.0f / sqrtf(32.f); a.ldx = ldx; a.ldp = ldp; a.ldy = ldy;
    a.groups = groups; a.G = group_tokens; a.per_wave = cdiv(group_tokens, 4); a.inner = inner;
    const int lds = inter ? RING_BYTES + XCH_BYTES : RING_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)dec_attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)dec_attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES + XCH_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    if (inter) hipLaunchKernelGGL(dec_attn_kernel<true>, dim3((unsigned)groups), dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dec_attn_kernel<false>, dim3((unsigned)cdiv(groups, 4)), dim3(256), lds, (hipStream_t)stream, a);
    return gom_launch_status();
}

/* The inter-instance block followed by the cross attention's offsets | logits product on its output (deformable_transformer.py:396-404
 * then ms_deform_attn.py:117-131's two Linear layers on query = tgt + query_pos): raw [rows, 384] = (Y + P2) Wraw^T + braw, one launch.
 * `image`: gom_dec_attn_image (inter = 1) + gom_dec_attn_raw_image. */
extern "C" int gom_dec_attn_raw_f32(const float* X, int ldx, const void* image, float eps, float* Y, int ldy, const float* P2, int ldp2,
                                    float* raw, int ldraw, int groups, int group_tokens, int inner, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && Y && P2 && raw && groups >= 0 && group_tokens > 0 && group_tokens <= 128 && inner > 0);
    GOM_CHECK_ARG(ldx >= D && ldy >= D && ldp2 >= D && ldraw >= 384 && (ldx % 4) == 0 && (ldy % 4) == 0 && (ldp2 % 4) == 0 && (ldraw % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)P2 % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)raw % 16) == 0 &&
                  ((uintptr_t)image % 16) == 0);
    if (groups == 0) return GOM_OK;
    DecArgs a{};
    a.X = X; a.P = nullptr; a.img = (const unsigned char*)image; a.Y = Y; a.flag = flag; a.eps = eps; a.scale = 1.0f / sqrtf(32.f);
    a.ldx = ldx; a.ldp = 0; a.ldy = ldy; a.groups = groups; a.G = group_tokens; a.per_wave = cdiv(group_tokens, 4); a.inner = inner;
    a.P2 = P2; a.ldp2 = ldp2; a.RAWO = raw; a.ldraw = ldraw;
    const int lds = RING_BYTES + XCH_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)dec_attn_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    hipLaunchKernelGGL((dec_attn_kernel<true, true>), dim3((unsigned)groups), dim3(256), lds, (hipStream_t)stream, a);
    return gom_launch_status();
}
