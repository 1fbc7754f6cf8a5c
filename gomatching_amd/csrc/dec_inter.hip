// Inter-instance self-attention of a DeepSolo composite decoder layer for MORE than 128 queries per frame (GoMatching++ /
// DSText: 300 queries, /root/reference/configs/GoMatching_PP_DSText.yaml; deformable_transformer.py:396-404):
//
//     attn[token, 32 h .. 32 h + 31] = softmax(q_h k_h^T / sqrt(32)) v_h      q | k | v = in_proj(tgt)      per (frame, point)
//
// csrc/dec_attn.hip keeps a whole attention group (<= 128 tokens) in the registers of ONE workgroup; 300 tokens do not fit
// there (a wave owns 32 tokens as 128 VGPRs of MFMA operand fragments), so the block splits differently:
//
//   * one workgroup = one (group, HEAD): 8 x more workgroups than groups (1 600 at 8 frames x 25 points: 6.25 rounds of the
//     chip), each needs only its head's 96 in_proj rows -- three 36 KB stages of dec_attn.hip's fragment-linear image;
//   * phase 1: the head's K and V^T of ALL tokens.  Wave w walks the token blocks w, w + 4, w + 8 (<= 32 tokens each): rows as
//     whole lines -> operand fragments through an LDS scratch (common.h), k transposed / v straight exactly as dec_attn.hip,
//     fp16 two-plane fragments into an 8 KB slot per block (the block's own slot doubles as its scratch);
//   * phase 2: the query weights replace the key weights in LDS, every wave walks its blocks again: q, then S^T = K . Q^T and
//     O^T += V^T . P^T over the key blocks with an ONLINE softmax (running max / sum per query = per lane), P straight from the
//     S^T accumulators after an fp16 split -- the registers never hold more than one key block's scores;
//   * O leaves as fp32 [token, 32 h ..]; out_proj + residual + LayerNorm are the existing proj_ln launch.
// The rows of a group are read twice per head (16 times per layer, L2-resident: 300 KB per group) instead of the q | k | v
// round trip through HBM ([rows, 768] written and read) and a VALU attention core.  f16x3 products (x-lo w-hi, x-hi w-lo,
// x-hi w-hi), fp32 accumulation and softmax; operands must stay within fp16's range (*flag otherwise, gemm_f16x3.hip's contract).
#include "common.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 256, NH = 8;
constexpr int FRAG = 1024;
constexpr int W_FRAGS = 32;
constexpr int CHUNK_FRAGS = W_FRAGS + 4;                 // dec_attn.hip's stage: 32 weight fragments + aux (1 / scale | bias) + 3 unused
constexpr int CHUNK_BYTES = CHUNK_FRAGS * FRAG;
constexpr int STAGES = 3 * NH + NH;
constexpr int IMAGE_BYTES = STAGES * CHUNK_BYTES;
constexpr int SLOT_BYTES = 8 * FRAG;                     // K (4 fragments) + V^T (4 fragments) of one token block
constexpr int MAX_BLOCKS = 11;                           // 2 x 36 KB + 11 x 8 KB = 160 KB of LDS

struct InterArgs {
    const float* X;
    const unsigned char* img;
    float* O;
    int* flag;
    float scale;
    int ldx, ldo;
    int G, inner, nblk, per_blk;
};

__device__ __forceinline__ void dma_fragment(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned frag_off, unsigned char* lds_frag) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_frag, 16, (int)lane_off, (int)frag_off, 0, 0);
}

__device__ __forceinline__ void acc_to_frags(const f32x16& a, half8 (&f)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        gom_split8_f16(f32x4{a[8 * s], a[8 * s + 1], a[8 * s + 2], a[8 * s + 3]},
                       f32x4{a[8 * s + 4], a[8 * s + 5], a[8 * s + 6], a[8 * s + 7]}, f[s][0], f[s][1]);
}

__device__ __forceinline__ f32x16 mfma_x3(const half8 a_hi, const half8 a_lo, const half8 b_hi, const half8 b_lo, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, c, 0, 0, 0);
    return c;
}

__global__ __launch_bounds__(256, 1) void dec_inter_heads_kernel(const InterArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wk = smem;                                // k weights, then q weights
    unsigned char* wv = smem + CHUNK_BYTES;                  // v weights, then the four waves' row scratch
    unsigned char* kv = smem + 2 * CHUNK_BYTES;              // per token block: K fragments 0..3, V^T fragments 4..7
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int head = blockIdx.x % NH;
    const long gi = blockIdx.x / NH;
    const long b = gi / p.inner, pp = gi % p.inner;

    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, IMAGE_BYTES, 0x00020000);
    const unsigned lane16 = lane * 16;
    // stage `st` of the image -> a 36 KB LDS region, the four waves sharing the 36 fragments
    auto request = [&](int st, unsigned char* dst) {
        for (int f = wave; f < CHUNK_FRAGS; f += 4) dma_fragment(rs_img, lane16, (unsigned)st * CHUNK_BYTES + f * FRAG, dst + f * FRAG);
    };
    request(3 * head + 1, wk);
    request(3 * head + 2, wv);

    float amax = 0.f, chk = 0.f;
    // token slot r (0..31) of block t -> its row; slots beyond the block's tokens re-read the block's first token (masked as keys,
    // never stored as queries)
    auto ntok = [&](int t) {
        const int left = p.G - t * p.per_blk;
        return left < 0 ? 0 : (left < p.per_blk ? left : p.per_blk);
    };
    auto row_of = [&](int t, int r) -> long {
        const long tq = (long)t * p.per_blk + (r < ntok(t) ? r : 0);
        return (b * p.G + tq) * p.inner + pp;
    };
    auto finish_t = [&](f32x16& acc, const float* aux) {     // transposed chunk: features (g & 3) + 8 (g >> 2) + 4 fh
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
    auto finish_s = [&](f32x16& acc, const float* aux) {     // straight chunk: the lane is the feature
        const float sc = aux[fr], bi = aux[32 + fr];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            acc[g] = fmaf(acc[g], sc, bi);
            amax = fmaxf(amax, fabsf(acc[g]));
        }
        asm volatile("" : "+v"(amax));
    };
    auto zero = [](f32x16& a) {
#pragma unroll
        for (int g = 0; g < 16; ++g) a[g] = 0.f;
    };

    // ---- phase 1: K and V^T of every token block of this head ----
    for (int t = wave; t < p.nblk; t += 4) {
        half8 xf[2][D / 16];
        float* scratch = reinterpret_cast<float*>(kv + t * SLOT_BYTES);      // the block's own slot: filled after the rows are split
        auto xrow = [&](int r) { return p.X + (size_t)row_of(t, r) * p.ldx; };
        gom_rows_to_fragments<64, false>(xrow, xrow, scratch, lane, xf, amax, [&]() {});
        if (t == wave) {                                     // the weights were requested in front of the first rows
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        f32x16 acc;
        zero(acc);
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            const half8 w_hi = *reinterpret_cast<const half8*>(wk + (2 * s) * FRAG + lane * 16);
            const half8 w_lo = *reinterpret_cast<const half8*>(wk + (2 * s + 1) * FRAG + lane * 16);
            acc = mfma_x3(w_hi, w_lo, xf[0][s], xf[1][s], acc);
        }
        finish_t(acc, reinterpret_cast<const float*>(wk + W_FRAGS * FRAG));
        half8 kf[2][2];
        acc_to_frags(acc, kf);
        zero(acc);
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            const half8 w_hi = *reinterpret_cast<const half8*>(wv + (2 * s) * FRAG + lane * 16);
            const half8 w_lo = *reinterpret_cast<const half8*>(wv + (2 * s + 1) * FRAG + lane * 16);
            acc = mfma_x3(xf[0][s], xf[1][s], w_hi, w_lo, acc);
        }
        finish_s(acc, reinterpret_cast<const float*>(wv + W_FRAGS * FRAG));
        half8 vf[2][2];
        acc_to_frags(acc, vf);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            *reinterpret_cast<half8*>(kv + t * SLOT_BYTES + f * FRAG + lane * 16) = kf[f >> 1][f & 1];
            *reinterpret_cast<half8*>(kv + t * SLOT_BYTES + (4 + f) * FRAG + lane * 16) = vf[f >> 1][f & 1];
        }
    }
    if (wave >= p.nblk) {                                    // a wave without a block still owns a share of the requests
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    __syncthreads();                                         // K / V^T complete, k and v weights dead
    request(3 * head, wk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- phase 2: q of every block, attention over all key blocks, online softmax ----
    for (int t = wave; t < p.nblk; t += 4) {
        half8 qf[2][2];
        {
            half8 xf[2][D / 16];
            float* scratch = reinterpret_cast<float*>(wv + wave * SLOT_BYTES);
            auto xrow = [&](int r) { return p.X + (size_t)row_of(t, r) * p.ldx; };
            gom_rows_to_fragments<64, false>(xrow, xrow, scratch, lane, xf, amax, [&]() {});
            f32x16 acc;
            zero(acc);
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                const half8 w_hi = *reinterpret_cast<const half8*>(wk + (2 * s) * FRAG + lane * 16);
                const half8 w_lo = *reinterpret_cast<const half8*>(wk + (2 * s + 1) * FRAG + lane * 16);
                acc = mfma_x3(w_hi, w_lo, xf[0][s], xf[1][s], acc);
            }
            finish_t(acc, reinterpret_cast<const float*>(wk + W_FRAGS * FRAG));
            acc_to_frags(acc, qf);
        }
        f32x16 o;
        zero(o);
        float m = -INFINITY, l = 0.f;                        // running max (whole query) and this half-wave's share of the sum
        for (int kb = 0; kb < p.nblk; ++kb) {
            const unsigned char* slot = kv + kb * SLOT_BYTES + lane * 16;
            const int nk = ntok(kb);
            f32x16 s;
            zero(s);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 k_hi = *reinterpret_cast<const half8*>(slot + (2 * ks) * FRAG);
                const half8 k_lo = *reinterpret_cast<const half8*>(slot + (2 * ks + 1) * FRAG);
                s = mfma_x3(k_hi, k_lo, qf[ks][0], qf[ks][1], s);
            }
            float mb = -INFINITY;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int j = (g & 3) + 8 * (g >> 2) + 4 * fh;   // key slot of register g
                s[g] = j < nk ? s[g] * p.scale : -INFINITY;
                mb = fmaxf(mb, s[g]);
            }
            mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
            const float m_new = fmaxf(m, mb);                // finite: every block holds at least one token
            const float alpha = __expf(m - m_new);           // 0 on the first block (m = -inf)
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                s[g] = __expf(s[g] - m_new);
                sum += s[g];
            }
            l = fmaf(l, alpha, sum);
            m = m_new;
#pragma unroll
            for (int g = 0; g < 16; ++g) o[g] *= alpha;
            half8 pf[2][2];
            acc_to_frags(s, pf);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 v_hi = *reinterpret_cast<const half8*>(slot + (4 + 2 * ks) * FRAG);
                const half8 v_lo = *reinterpret_cast<const half8*>(slot + (4 + 2 * ks + 1) * FRAG);
                o = mfma_x3(v_hi, v_lo, pf[ks][0], pf[ks][1], o);
            }
        }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        // O^T: lane = query token, registers = features (g & 3) + 8 (g >> 2) + 4 fh of head `head`
        const bool valid = fr < ntok(t);
        float* orow = p.O + (size_t)row_of(t, fr) * p.ldo + 32 * head + 4 * fh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = o[4 * q + e] * inv;
                chk = fmaf(v[e], 0.f, chk);
            }
            if (valid) *reinterpret_cast<f32x4*>(orow + 8 * q) = v;
        }
        asm volatile("" : "+v"(chk));
    }
    if ((!(amax <= 65504.f) || !(chk == 0.f)) && p.flag) atomicOr(p.flag, 1);
}

}  // namespace

extern "C" int gom_dec_inter_heads_f32(const float* X, int ldx, const void* image, float* O, int ldo, int groups,
                                       int group_tokens, int inner, int* flag, void* stream) {
    GOM_CHECK_ARG(X && image && O && groups >= 0 && group_tokens > 0 && inner > 0);
    GOM_CHECK_ARG(ldx >= D && ldo >= D && (ldx % 4) == 0 && (ldo % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)O % 16) == 0 && ((uintptr_t)image % 16) == 0);
    const int nblk = cdiv(group_tokens, 32);
    GOM_CHECK_ARG(nblk <= MAX_BLOCKS);                       // <= 352 tokens per group
    if (groups == 0) return GOM_OK;
    InterArgs a{};
    a.X = X; a.img = (const unsigned char*)image; a.O = O; a.flag = flag; a.scale = 1.0f / sqrtf(32.f);
    a.ldx = ldx; a.ldo = ldo; a.G = group_tokens; a.inner = inner; a.nblk = nblk; a.per_blk = cdiv(group_tokens, nblk);
    const int lds = 2 * CHUNK_BYTES + nblk * SLOT_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)dec_inter_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       2 * CHUNK_BYTES + MAX_BLOCKS * SLOT_BYTES);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    hipLaunchKernelGGL(dec_inter_heads_kernel, dim3((unsigned)groups * NH), dim3(256), lds, (hipStream_t)stream, a);
    return gom_launch_status();
}
