// One association match as ONE persistent kernel (lstmatcher.py:333-381, transformer.py:60-96,
// gom_lstmatcher.py:429-445 / 510-547): gather -> [encoder layer] -> decoder layer -> q.k^T -> per-frame softmax with the
// background logit -> trajectory score, with grid-wide barriers between the phases instead of kernel boundaries.
//
// Why: the problems are tiny (tens to a few hundred rows of d = 1024 against ~34 MB of weights) and the tracker is a
// serial per-frame recurrence; issued as 18 dependent kernels a match costs ~23 us of dependent-launch latency per kernel
// (~420 us, tools/match_breakdown.py), which -- replicated over the 8N frames of an N-GPU step -- bounds multi-GPU scaling
// (DESIGN.md 6).  Here a match is one launch of G resident workgroups; a phase hands over to the next through an atomic
// arrive-and-spin barrier (release / acquire fences at agent scope, so a phase sees what the other CUs wrote).
//
// Arithmetic: every linear layer uses the one-wave-per-column-times-8-rows fp32 FMA scheme of gemm_small.hip (an output
// depends only on its row, its column and K), attention is a wave per (query, head) with an exact two-pass softmax.
#include "common.h"

namespace {

constexpr int RM = 8;                 // rows of A per wave in the linear phases
constexpr int MAX_KEYS = 640;         // keys per attention problem (6 frames x 100 queries + slack): 10 per lane
constexpr int KPL = MAX_KEYS / 64;

struct FusedArgs {
    const float* pool;
    const int* rows;
    const int* offs;                  // frame offsets [T+1]
    const int* meta;                  // nonk[Np] | col_of[Np] | last_idx[M] | k_inds[n_k]
    const float* boxes;
    const float* decay;               // or nullptr
    gom_matcher_layer enc[2], dec[2];
    int n_enc, n_dec;
    int N, T, lo, n_k, num_tracks, d, heads, ffn, with_iou;
    float img_w, img_h, max_center_dist;
    float* ws;                        // workspace (floats), carved below
    unsigned* barrier;                // one counter, zero at launch
    float* traj;
};

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every storing wave drains its stores before the barrier
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                         // release: this workgroup's writes are visible device-wide
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the write-back completes before the arrival is published
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __threadfence();                                             // acquire: drop what this CU cached of the other CUs' buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ... and wait for the invalidate before the next phase loads
}

// C[M, Nout] = act(A[M, K] . W[Nout, K]^T + bias + R): a wave owns CB columns x 8 rows per item -- the per-output arithmetic
// of gemm_small_kernel (lane-strided fp32 FMA chains + a fixed butterfly), but CB weight rows stream at once and the 8
// activation rows are loaded once for all of them, so that a persistent wave has 12 independent 16-byte loads in flight
// per step instead of 9 (of which 8 hit the same few cache lines)
constexpr int CB = 4;
__device__ void linear_phase(const float* A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                             const float* R, int ldr, int relu, float* C, int ldc, int M, int Nout, int K, int gw, int GW,
                             int lane) {
    const int groups = (M + RM - 1) / RM, ncb = (Nout + CB - 1) / CB;
    for (long item = gw; item < (long)groups * ncb; item += GW) {
        const int n0 = (int)(item % ncb) * CB, m0 = (int)(item / ncb) * RM;
        const float* w[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) w[c] = W + (size_t)(n0 + c < Nout ? n0 + c : Nout - 1) * ldw;
        const float* a[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) a[r] = A + (size_t)(m0 + r < M ? m0 + r : M - 1) * lda;
        float acc[CB][RM];
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int r = 0; r < RM; ++r) acc[c][r] = 0.f;
#pragma unroll 2
        for (int k = lane * 4; k < K; k += 256) {
            f32x4 y[CB], x[RM];
#pragma unroll
            for (int c = 0; c < CB; ++c) y[c] = *reinterpret_cast<const f32x4*>(w[c] + k);
#pragma unroll
            for (int r = 0; r < RM; ++r) x[r] = *reinterpret_cast<const f32x4*>(a[r] + k);
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int r = 0; r < RM; ++r) {
                    acc[c][r] = fmaf(x[r][0], y[c][0], acc[c][r]);
                    acc[c][r] = fmaf(x[r][1], y[c][1], acc[c][r]);
                    acc[c][r] = fmaf(x[r][2], y[c][2], acc[c][r]);
                    acc[c][r] = fmaf(x[r][3], y[c][3], acc[c][r]);
                }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int r = 0; r < RM; ++r) acc[c][r] = wave_sum(acc[c][r]);
        if (lane < CB && n0 + lane < Nout) {                      // lane c stores column n0 + c
            const int n = n0 + lane;
            const float sh = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const int m = m0 + r;
                if (m < M) {
                    float v = (lane == 0 ? acc[0][r] : lane == 1 ? acc[1][r] : lane == 2 ? acc[2][r] : acc[3][r]) + sh;
                    if (R) v += R[(size_t)m * ldr + n];
                    C[(size_t)m * ldc + n] = relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
}

// softmax(q k^T / sqrt(hd)) v for hd = 128: one wave per (query, head); lane owns keys lane, lane+64, ... and output
// dimensions 2 lane, 2 lane + 1
__device__ void attention_phase(const float* q, int ld_q, const float* k, const float* v, int ld_kv, float* o, int ld_o,
                                int Lq, int Lk, int heads, int gw, int GW, int lane) {
    constexpr int HD = 128;
    const float scale = 0.08838834764831845f;                   // 1 / sqrt(128)
    for (long item = gw; item < (long)Lq * heads; item += GW) {
        const int i = (int)(item / heads), h = (int)(item % heads);
        const float* qr = q + (size_t)i * ld_q + h * HD;
        float s[KPL];
#pragma unroll
        for (int c = 0; c < KPL; ++c) s[c] = 0.f;
        for (int d0 = 0; d0 < HD; d0 += 4) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(qr + d0) * scale;
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const int j = c * 64 + lane;
                if (j < Lk) {
                    const f32x4 kv = *reinterpret_cast<const f32x4*>(k + (size_t)j * ld_kv + h * HD + d0);
                    s[c] = fmaf(qv[0], kv[0], s[c]); s[c] = fmaf(qv[1], kv[1], s[c]);
                    s[c] = fmaf(qv[2], kv[2], s[c]); s[c] = fmaf(qv[3], kv[3], s[c]);
                }
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < KPL; ++c)
            if (c * 64 + lane < Lk) mx = fmaxf(mx, s[c]);
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            s[c] = (c * 64 + lane < Lk) ? expf(s[c] - mx) : 0.f;
            sum += s[c];
        }
        const float inv = 1.f / wave_sum(sum);
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int jend = min(Lk - c * 64, 64);
            for (int jj = 0; jj < jend; ++jj) {
                const float p = __shfl(s[c], jj, 64);
                const float* vr = v + (size_t)(c * 64 + jj) * ld_kv + h * HD + 2 * lane;
                o0 = fmaf(p, vr[0], o0);
                o1 = fmaf(p, vr[1], o1);
            }
        }
        float* orow = o + (size_t)i * ld_o + h * HD + 2 * lane;
        orow[0] = o0 * inv;
        orow[1] = o1 * inv;
    }
}

constexpr int FW = 4;                 // waves per workgroup (8 measured slower beside a saturated GPU: 0.83 vs 0.62 ms per match)
__global__ __launch_bounds__(64 * FW) void match_fused_kernel(const FusedArgs p) {
    __shared__ float act_s[FW][MAX_KEYS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = gridDim.x, GW = G * FW, gw = blockIdx.x * FW + wave;
    const int N = p.N, n_k = p.n_k, d = p.d, ffn = p.ffn;
    const long wide = 3L * d > ffn ? 3L * d : ffn;
    float* w0 = p.ws;
    float* src = w0;         w0 += (long)N * d;
    float* mem_a = w0;       w0 += (long)N * d;
    float* mem_b = w0;       w0 += (long)N * d;
    float* big = w0;         w0 += (long)N * wide;
    float* att = w0;         w0 += (long)N * d;
    float* tgt_a = w0;       w0 += (long)n_k * d;
    float* tgt_b = w0;       w0 += (long)n_k * d;
    float* qbuf = w0;        w0 += (long)n_k * d;
    float* hid = w0;         w0 += (long)n_k * wide;
    float* logits = w0;
    unsigned phase = 0;
#define NEXT_PHASE() grid_barrier(p.barrier, (++phase) * (unsigned)G)

    // ---- gather
    for (long i = (long)blockIdx.x * (64 * FW) + threadIdx.x; i < (long)N * (d / 4); i += (long)G * (64 * FW)) {
        const int r = (int)(i / (d / 4)), c = (int)(i % (d / 4));
        *reinterpret_cast<f32x4*>(src + i * 4) = *reinterpret_cast<const f32x4*>(p.pool + ((size_t)p.rows[r] * (d / 4) + c) * 4);
    }
    NEXT_PHASE();
    const float* memory = src;
    for (int l = 0; l < p.n_enc; ++l) {                        // post-norm layer with Identity norms (transformer.py:180-195)
        const gom_matcher_layer& L = p.enc[l];
        float* out = mem_b;                                      // one encoder layer at most (checked by the host): never `src`
        linear_phase(memory, d, L.in_w, d, L.in_b, nullptr, 0, 0, big, 3 * d, N, 3 * d, d, gw, GW, lane);
        NEXT_PHASE();
        attention_phase(big, 3 * d, big + d, big + 2 * d, 3 * d, att, d, N, N, p.heads, gw, GW, lane);
        NEXT_PHASE();
        linear_phase(att, d, L.out_w, d, L.out_b, memory, d, 0, mem_a, d, N, d, d, gw, GW, lane);
        NEXT_PHASE();
        linear_phase(mem_a, d, L.lin1_w, d, L.lin1_b, nullptr, 0, 1, big, ffn, N, ffn, d, gw, GW, lane);
        NEXT_PHASE();
        linear_phase(big, ffn, L.lin2_w, ffn, L.lin2_b, mem_a, d, 0, out, d, N, d, ffn, gw, GW, lane);
        NEXT_PHASE();
        memory = out;
    }
    const float* tgt = src + (long)p.lo * d;                     // tgt = src[query rows] (transformer.py:80-84)
    for (int l = 0; l < p.n_dec; ++l) {                        // cross-attention only (transformer.py:270-294)
        const gom_matcher_layer& L = p.dec[l];
        linear_phase(tgt, d, L.in_w, d, L.in_b, nullptr, 0, 0, qbuf, d, n_k, d, d, gw, GW, lane);
        linear_phase(memory, d, L.in_w + (size_t)d * d, d, L.in_b ? L.in_b + d : nullptr, nullptr, 0, 0, big, 2 * d, N, 2 * d, d,
                     gw, GW, lane);
        NEXT_PHASE();
        attention_phase(qbuf, d, big, big + d, 2 * d, att, d, n_k, N, p.heads, gw, GW, lane);
        NEXT_PHASE();
        float* out = (tgt == tgt_a) ? tgt_b : tgt_a;
        linear_phase(att, d, L.out_w, d, L.out_b, tgt, d, 0, out, d, n_k, d, d, gw, GW, lane);
        NEXT_PHASE();
        tgt = out;
        if (L.lin1_w) {
            float* out2 = (tgt == tgt_a) ? tgt_b : tgt_a;
            linear_phase(tgt, d, L.lin1_w, d, L.lin1_b, nullptr, 0, 1, hid, ffn, n_k, ffn, d, gw, GW, lane);
            NEXT_PHASE();
            linear_phase(hid, ffn, L.lin2_w, ffn, L.lin2_b, tgt, d, 0, out2, d, n_k, d, ffn, gw, GW, lane);
            NEXT_PHASE();
            tgt = out2;
        }
    }
    // ---- ATTWeightHead with 0 layers: q . k^T (lstmatcher.py:360-371)
    linear_phase(tgt, d, memory, d, nullptr, nullptr, 0, 0, logits, N, n_k, N, d, gw, GW, lane);
    NEXT_PHASE();
    // ---- per-frame softmax with the appended zero logit (lstmatcher.py:373-381) + trajectory score: a wave per query row
    const int Np = N - n_k, M = p.num_tracks;
    const int* nonk = p.meta;
    const int* col_of = p.meta + Np;
    const int* last_idx = p.meta + 2 * Np;
    const int* k_inds = p.meta + 2 * Np + M;
    for (int i = gw; i < n_k; i += GW) {
        const float* row = logits + (size_t)i * N;
        float* act = act_s[wave];
        for (int t = 0; t < p.T; ++t) {
            const int lo = p.offs[t], hi = p.offs[t + 1];
            float mx = 0.f;                                      // the appended background logit
            for (int j = lo + lane; j < hi; j += 64) mx = fmaxf(mx, row[j]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = lo + lane; j < hi; j += 64) sum += expf(row[j] - mx);
            sum = wave_sum(sum) + expf(0.f - mx);
            for (int j = lo + lane; j < hi; j += 64) act[j] = expf(row[j] - mx) / sum;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // the row is wave-private: LDS writes before the reads below
        __builtin_amdgcn_wave_barrier();
        const float* kb = p.boxes + (size_t)k_inds[i] * 4;
        const float kx0 = kb[0] / p.img_w, ky0 = kb[1] / p.img_h, kx1 = kb[2] / p.img_w, ky1 = kb[3] / p.img_h;
        const float kcx = (kx0 + kx1) / 2.f, kcy = (ky0 + ky1) / 2.f;
        const float ks = (kx1 - kx0) * (kx1 - kx0) + (ky1 - ky0) * (ky1 - ky0);
        for (int m = lane; m < M; m += 64) {
            float s = 0.f;
            bool any_valid = false;
            for (int j = 0; j < Np; ++j) {
                if (col_of[j] != m) continue;
                float a = act[nonk[j]];
                if (p.decay) a *= p.decay[j];
                s += a;
                if (p.max_center_dist > 0.f) {
                    const float* nb = p.boxes + (size_t)nonk[j] * 4;
                    const float nx0 = nb[0] / p.img_w, ny0 = nb[1] / p.img_h, nx1 = nb[2] / p.img_w, ny1 = nb[3] / p.img_h;
                    const float dx = kcx - (nx0 + nx1) / 2.f, dy = kcy - (ny0 + ny1) / 2.f;
                    if ((dx * dx + dy * dy) / (ks + 1e-8f) < p.max_center_dist) any_valid = true;
                }
            }
            if (p.with_iou) {
                const float* lb = p.boxes + (size_t)nonk[last_idx[m]] * 4;
                const float lx0 = lb[0] / p.img_w, ly0 = lb[1] / p.img_h, lx1 = lb[2] / p.img_w, ly1 = lb[3] / p.img_h;
                const float w = fmaxf(fminf(kx1, lx1) - fmaxf(kx0, lx0), 0.f);
                const float hh = fmaxf(fminf(ky1, ly1) - fmaxf(ky0, ly0), 0.f);
                const float inter = w * hh;
                const float a1 = (kx1 - kx0) * (ky1 - ky0), a2 = (lx1 - lx0) * (ly1 - ly0);
                const float iou = inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
                s = fmaxf(s, iou);
            }
            if (p.max_center_dist > 0.f && !any_valid) s = 0.f;
            p.traj[(size_t)i * M + m] = s;
        }
        __builtin_amdgcn_wave_barrier();                         // the next row of this wave overwrites act
    }
#undef NEXT_PHASE
}

}  // namespace

// Workgroups of the persistent grid (all of them have to be resident before the first barrier opens).  Measured beside a
// saturated GPU (tools/match_breakdown.py 8 <G>), one column per wave: 4 -> 8.0 ms per match, 8 -> 4.1, 16 -> 2.2, 32 -> 1.27,
// 64 -> 0.84; with CB = 4 columns per wave: 32 -> 0.83, 64 -> 0.62, 128 -> 0.63.  The kernel is bound by its own per-wave
// latency, not by waiting for slots; the 18-kernel chain takes 0.43 ms, so this form stays optional (ops.FUSED_MATCHER).
static int g_fused_grid = 64;
extern "C" int gom_match_fused_set_grid(int workgroups) {
    if (workgroups < 1 || workgroups > 256) return GOM_ERR_INVALID_ARG;
    g_fused_grid = workgroups;
    return GOM_OK;
}

// Largest problem the fused form takes (beyond it the per-kernel chain with its MFMA GEMMs is the better tool).
extern "C" int gom_match_fused_supported(int N, int n_k, int d, int heads, int n_enc, int n_dec) {
    return N > 0 && N <= 256 && n_k > 0 && n_k <= N && d % 256 == 0 && heads > 0 && d / heads == 128 && n_enc >= 0 && n_enc <= 1 &&
           n_dec >= 0 && n_dec <= 2;
}

/* Same contract as gom_match_scores_f32; workspace: gom_match_workspace_floats(...) floats, its LAST 64 floats are used as
 * the barrier word (zeroed here with a memset node on the stream). */
extern "C" int gom_match_scores_fused_f32(const float* pool, int ld_pool, const int* rows, const int* frame_offsets,
                                          const int* meta, const float* boxes, const float* decay, int N, int T, int lo,
                                          int hi, int num_tracks, const gom_matcher_layer* enc, int n_enc,
                                          const gom_matcher_layer* dec, int n_dec, int d, int heads, int ffn, float img_w,
                                          float img_h, int with_iou, float max_center_dist, float* workspace,
                                          long workspace_floats, float* traj, void* stream) {
    if (!pool || !rows || !frame_offsets || !meta || !boxes || !workspace || !traj) return GOM_ERR_INVALID_ARG;
    if (T <= 0 || lo < 0 || hi <= lo || hi > N || num_tracks <= 0 || ld_pool != d) return GOM_ERR_INVALID_ARG;
    const int n_k = hi - lo;
    if (!gom_match_fused_supported(N, n_k, d, heads, n_enc, n_dec)) return GOM_ERR_UNSUPPORTED;
    if ((n_enc > 0 && !enc) || (n_dec > 0 && !dec)) return GOM_ERR_INVALID_ARG;
    const long need = gom_match_workspace_floats(N, n_k, d, ffn);
    if (need < 0 || workspace_floats < need) return GOM_ERR_INVALID_ARG;
    FusedArgs a{};
    a.pool = pool; a.rows = rows; a.offs = frame_offsets; a.meta = meta; a.boxes = boxes; a.decay = decay;
    for (int l = 0; l < n_enc; ++l) {
        a.enc[l] = enc[l];
        if (!enc[l].in_w || !enc[l].out_w || !enc[l].lin1_w || !enc[l].lin2_w) return GOM_ERR_INVALID_ARG;
    }
    for (int l = 0; l < n_dec; ++l) {
        a.dec[l] = dec[l];
        if (!dec[l].in_w || !dec[l].out_w || (dec[l].lin1_w && !dec[l].lin2_w)) return GOM_ERR_INVALID_ARG;
    }
    a.n_enc = n_enc; a.n_dec = n_dec; a.N = N; a.T = T; a.lo = lo; a.n_k = n_k; a.num_tracks = num_tracks; a.d = d;
    a.heads = heads; a.ffn = ffn; a.with_iou = with_iou; a.img_w = img_w; a.img_h = img_h; a.max_center_dist = max_center_dist;
    a.ws = workspace; a.traj = traj;
    a.barrier = reinterpret_cast<unsigned*>(workspace + (workspace_floats - 64));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(a.barrier, 0, 64 * sizeof(float), s);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    const int G = g_fused_grid;
    hipLaunchKernelGGL(match_fused_kernel, dim3(G), dim3(64 * FW), 0, s, a);
    return gom_launch_status();
}
