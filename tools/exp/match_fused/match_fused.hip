// The device chain of ONE long-term match (matcher_rt.cpp: gather, the encoder layer, the cross-attention decoder layer, the
// association logits, per-frame softmax and trajectory scores -- lstmatcher.py:333-381, transformer.py:60-96,
// gom_lstmatcher.py:429-445/510-547) as ONE launch.
//
// Why it was built: the replicated tracker of a multi-GPU step runs this chain once per frame with unmatched detections (~45 of the
// 64 frames of an 8-GPU step), 13 dependent launches each, and with the chain cut to 3 launches the 8-GPU-load step drops from
// 33.9-35.4 ms to 30.5 ms (profiles/r05_tracker_chain_skip_experiment.log: launches AND work removed).
// What it measures (profiles/r05_match_fused_ab.log), and why the tracker does NOT use it by default (gom_tracker_set_fused):
//   * alone, a match of 49 window rows takes 132 us as the chain and 324 / 246 / 232 us as one launch of 32 / 64 / 128 workgroups:
//     a grid barrier costs what a launch boundary costs (8-14 us: L2 write-back, arrival atomic, poll, invalidate = four dependent
//     trips to memory), and between barriers 32 workgroups run a phase's 896 wave tasks in 3.5 rounds where the chain's launch
//     spreads them over the whole chip in one;
//   * beside the detector with 8 GPUs' tracker load: 34.5 ms per step against 33.9 on the 32-CU lane (the step is bound by the
//     detector's 224 CUs there, not by the tracker), 36.1 against 34.6-36.7 without a lane -- where its 252-register workgroups need
//     EMPTY CUs and one run in three hit the barrier's 2 s time-out (a workgroup that is not resident in time).
// It stays as a correct, bit-identical form for A/B runs (bench.py --match-fused) and for its parity tests.
//
// How: a small persistent grid (G workgroups of 512 threads, all co-resident) walks the chain's phases; a phase is a grid-stride
// loop over the SAME wave tasks the chain's kernels run (tracker_tasks.h: 8 x 8 output patches of the skinny GEMMs, one
// (head, query row) of the tiny attention, one current detection of the score), and a grid barrier separates dependent phases.
// An output's arithmetic is a function of its task alone, so the result is the chain's BIT FOR BIT (tests/test_match_fused_gpu.py);
// the chain stays as the path for everything this kernel does not take (more than 64 rows in the window, head_dim != 128, no
// hoisted projections).
//
// Grid barrier: count + generation words (caller-owned, zero once; the count is back at zero when the kernel ends).  The last
// workgroup to arrive resets the count and bumps the generation (release, agent scope: L2 write-back on this XCD); the others
// spin on the generation (acquire: L1 / non-local L2 lines invalidated) -- the eight XCDs' L2s are not coherent with each other
// for plain loads, only through these scoped operations.  Every workgroup must be resident: the launcher never asks for more
// workgroups than the stream has CUs (`gom_match_fused_set_grid`), and a workgroup fits an empty CU by construction.  A barrier
// that is not passed within ~2 s (100 MHz s_memrealtime) gives up and raises the status word: an error for the caller, never a
// hung GPU.
#include "common.h"
#include "tracker_tasks.h"

namespace {

constexpr int THREADS = 512;
constexpr int MAX_LAYERS = 4;

struct FusedArgs {
    const float* pool; int ld_pool;
    const float* proj; int ld_proj;
    const int* rows; const int* frame_offsets; const int* meta; const float* boxes; const float* decay;
    int N, T, lo, n_k, num_tracks;
    gom_matcher_layer enc[MAX_LAYERS]; int n_enc;
    gom_matcher_layer dec[MAX_LAYERS]; int n_dec;
    int d, heads, ffn;
    float img_w, img_h; int with_iou; float max_center_dist;
    float scale;                                              // 1 / sqrt(head_dim), formed on the host as gom_mha_core_f32 forms it
    // workspace (matcher_rt.cpp's layout)
    float *src, *mem_a, *mem_b, *mem_c, *big, *att, *tgt_a, *tgt_b, *qbuf, *hid, *logits;
    float* traj;
    unsigned* sync;                                           // [0] arrivals, [1] generation
    int* status;                                              // != 0: a barrier timed out (results invalid)
    const unsigned* desc_src; unsigned* desc_dst; long desc_words;   // optional upload of the descriptors (phase 0)
};

struct Grid {
    unsigned* sync;
    int* status;
    unsigned gen;                                             // generation this workgroup waits to leave
    bool dead;
};

__device__ __forceinline__ void grid_barrier(Grid& g) {
    __syncthreads();                                          // (workgroup-scope release: every wave's stores have left for L2)
    if (threadIdx.x == 0 && !g.dead) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // write back this XCD's L2
        const unsigned arrived = __hip_atomic_fetch_add(g.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            __hip_atomic_store(g.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.sync + 1, g.gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(g.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.gen) {
                __builtin_amdgcn_s_sleep(4);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {   // ~2 s: a workgroup never became resident
                    g.dead = true;
                    __hip_atomic_store(g.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drop stale L1 / L2 lines before the next phase reads
    }
    ++g.gen;
    __syncthreads();
}

__device__ __forceinline__ void linear(const float* A, int lda, int M, const float* W, const float* b, int N, int K, const float* R,
                                       int ldr, int relu, float* C, int ldc, long gwave, long nwaves, int lane) {
    if (M <= 0) return;
    const long tasks = gom_tasks::gemm_small_tasks(M, N);
    for (long o = gwave; o < tasks; o += nwaves)
        gom_tasks::gemm_small_task(A, nullptr, lda, W, K, nullptr, b, R, ldr, relu, C, ldc, M, N, K, o, lane);
}

__device__ __forceinline__ void attend(const float* q, int ld_q, const float* k, const float* v, int ld_kv, float* o, int d,
                                       int heads, int Lq, int Lk, float scale, long gwave, long nwaves, int lane) {
    const long tasks = (long)heads * Lq;
    for (long w = gwave; w < tasks; w += nwaves)
        gom_tasks::mha_tiny128_task(q, k, v, o, Lq, Lk, 1, heads, 0, 0, ld_q, 0, 0, ld_kv, 0, 0, ld_kv, 0, 0, d, scale, w, lane);
}

__global__ __launch_bounds__(THREADS, 1) void match_fused_kernel(const FusedArgs p) {
    extern __shared__ float act[];                            // [N] activations of one current detection
    const int tid = threadIdx.x, lane = tid & 63;
    const long nwaves = (long)gridDim.x * (THREADS / 64), gwave = (long)blockIdx.x * (THREADS / 64) + (tid >> 6);
    Grid g{p.sync, p.status, 0u, false};
    if (tid == 0) g.gen = __hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int N = p.N, n_k = p.n_k, d = p.d, ffn = p.ffn;
    const float scale = p.scale;

    if (p.desc_src) {   // the match's descriptors (rows | offsets | meta | boxes | decay) from pinned host memory: one launch fewer
        const long nthreads = (long)gridDim.x * THREADS;
        for (long i = (long)blockIdx.x * THREADS + tid; i < p.desc_words; i += nthreads) p.desc_dst[i] = p.desc_src[i];
        grid_barrier(g);
    }
    {   // the window's embeddings, their encoder-layer-0 in-projections, the current frame's decoder-layer-0 query projections
        const int d4 = d / 4;
        const long items = ((long)N * 4 + n_k) * d4, nthreads = (long)gridDim.x * THREADS;
        for (long i = (long)blockIdx.x * THREADS + tid; i < items; i += nthreads)
            gom_tasks::gather_match_item(p.pool, p.ld_pool, p.proj, p.ld_proj, p.rows, N, p.lo, n_k, d4, p.src, p.big, p.qbuf, i);
    }
    grid_barrier(g);
    const float* memory = p.src;
    for (int l = 0; l < p.n_enc; ++l) {                       // post-norm layer with Identity norms (transformer.py:180-195)
        const gom_matcher_layer& L = p.enc[l];
        float* out = (memory == p.mem_b) ? p.mem_c : p.mem_b;
        if (l > 0) {
            linear(memory, d, N, L.in_w, L.in_b, 3 * d, d, nullptr, 0, 0, p.big, 3 * d, gwave, nwaves, lane);
            grid_barrier(g);
        }
        attend(p.big, 3 * d, p.big + d, p.big + 2 * d, 3 * d, p.att, d, p.heads, N, N, scale, gwave, nwaves, lane);
        grid_barrier(g);
        linear(p.att, d, N, L.out_w, L.out_b, d, d, memory, d, 0, p.mem_a, d, gwave, nwaves, lane);
        grid_barrier(g);
        linear(p.mem_a, d, N, L.lin1_w, L.lin1_b, ffn, d, nullptr, 0, 1, p.big, ffn, gwave, nwaves, lane);
        grid_barrier(g);
        linear(p.big, ffn, N, L.lin2_w, L.lin2_b, d, ffn, p.mem_a, d, 0, out, d, gwave, nwaves, lane);
        grid_barrier(g);
        memory = out;
    }
    const float* tgt = p.src + (long)p.lo * d;                // tgt = src[query rows] (transformer.py:80-84)
    for (int l = 0; l < p.n_dec; ++l) {                       // cross-attention only (transformer.py:270-294)
        const gom_matcher_layer& L = p.dec[l];
        if (l > 0) linear(tgt, d, n_k, L.in_w, L.in_b, d, d, nullptr, 0, 0, p.qbuf, d, gwave, nwaves, lane);
        linear(memory, d, N, L.in_w + (size_t)d * d, L.in_b ? L.in_b + d : nullptr, 2 * d, d, nullptr, 0, 0, p.big, 2 * d, gwave,
               nwaves, lane);
        grid_barrier(g);
        attend(p.qbuf, d, p.big, p.big + d, 2 * d, p.att, d, p.heads, n_k, N, scale, gwave, nwaves, lane);
        grid_barrier(g);
        float* out = (tgt == p.tgt_a) ? p.tgt_b : p.tgt_a;
        linear(p.att, d, n_k, L.out_w, L.out_b, d, d, tgt, d, 0, out, d, gwave, nwaves, lane);
        grid_barrier(g);
        tgt = out;
        if (L.lin1_w) {
            float* out2 = (tgt == p.tgt_a) ? p.tgt_b : p.tgt_a;
            linear(tgt, d, n_k, L.lin1_w, L.lin1_b, ffn, d, nullptr, 0, 1, p.hid, ffn, gwave, nwaves, lane);
            grid_barrier(g);
            linear(p.hid, ffn, n_k, L.lin2_w, L.lin2_b, d, ffn, tgt, d, 0, out2, d, gwave, nwaves, lane);
            grid_barrier(g);
            tgt = out2;
        }
    }
    // ATTWeightHead with 0 layers: q . k^T (lstmatcher.py:360-371)
    linear(tgt, d, n_k, memory, nullptr, N, d, nullptr, 0, 0, p.logits, N, gwave, nwaves, lane);
    grid_barrier(g);
    for (int i = blockIdx.x; i < n_k; i += gridDim.x)
        gom_tasks::asso_score_block(p.logits, N, p.frame_offsets, p.T, p.meta, p.decay, p.boxes, p.img_w, p.img_h, N - n_k,
                                    p.num_tracks, p.with_iou, p.max_center_dist, p.traj, i, act, tid, THREADS);
}

int g_grid = 32;

}  // namespace

/* [host] workgroups of the fused match kernel (default 32).  Every one of them must be resident at the same time: never more than
 * the CUs of the stream the match runs on (GoMatching.reserve_tracker_cus sets it with the lane's size). */
extern "C" int gom_match_fused_set_grid(int workgroups) {
    if (workgroups < 1 || workgroups > 256) return GOM_ERR_INVALID_ARG;
    g_grid = workgroups;
    return GOM_OK;
}

/* [host] 1: this match runs as one launch (gom_match_fused_f32), 0: it needs the chain (gom_match_scores_proj_f32). */
extern "C" int gom_match_fused_serves(int N, int n_k, int n_enc, int n_dec, int d, int heads, int ffn, int has_proj) {
    return has_proj && N >= 1 && N <= 64 && n_k >= 1 && n_k <= N && n_enc >= 0 && n_enc <= MAX_LAYERS && n_dec >= 0 &&
                   n_dec <= MAX_LAYERS && d > 0 && heads > 0 && d == heads * 128 && ffn > 0 && (ffn % 4) == 0
               ? 1
               : 0;
}

/* gom_match_scores_proj_f32 as ONE launch, same bits (see the top of this file).  `sync` [device, 2 words]: the grid barrier's
 * state, zero when first used, owned by the caller and by one match at a time; `status` [device-visible int, e.g. pinned host
 * memory]: set to 1 if a barrier gave up -- the caller clears it before the call and reads it after the stream is done.
 * desc_host != NULL: the kernel first copies `desc_words` 32-bit words from desc_host (device-visible host memory) to desc_dev,
 * the block that rows / frame_offsets / meta / boxes / decay point into (the tracker's descriptor upload without its own launch). */
extern "C" int gom_match_fused_f32(const float* pool, int ld_pool, const float* proj, int ld_proj, const int* rows,
                                   const int* frame_offsets, const int* meta, const float* boxes, const float* decay, int N,
                                   int T, int lo, int hi, int num_tracks, const gom_matcher_layer* enc, int n_enc,
                                   const gom_matcher_layer* dec, int n_dec, int d, int heads, int ffn, float img_w, float img_h,
                                   int with_iou, float max_center_dist, float* workspace, long workspace_floats, float* traj,
                                   unsigned int* sync, int* status, const void* desc_host, void* desc_dev, long desc_words,
                                   void* stream) {
    if (!pool || !proj || !rows || !frame_offsets || !meta || !boxes || !workspace || !traj || !sync || !status)
        return GOM_ERR_INVALID_ARG;
    if (N <= 0 || T <= 0 || lo < 0 || hi <= lo || hi > N || num_tracks <= 0) return GOM_ERR_INVALID_ARG;
    if ((n_enc > 0 && !enc) || (n_dec > 0 && !dec)) return GOM_ERR_INVALID_ARG;
    const int n_k = hi - lo;
    if (!gom_match_fused_serves(N, n_k, n_enc, n_dec, d, heads, ffn, 1)) return GOM_ERR_UNSUPPORTED;
    if (ld_proj < 4 * d || ld_pool < d || (ld_pool % 4) || (ld_proj % 4)) return GOM_ERR_INVALID_ARG;
    if (workspace_floats < gom_match_workspace_floats(N, n_k, d, ffn)) return GOM_ERR_INVALID_ARG;
    FusedArgs a{};
    a.pool = pool; a.ld_pool = ld_pool; a.proj = proj; a.ld_proj = ld_proj;
    a.rows = rows; a.frame_offsets = frame_offsets; a.meta = meta; a.boxes = boxes; a.decay = decay;
    a.N = N; a.T = T; a.lo = lo; a.n_k = n_k; a.num_tracks = num_tracks;
    for (int l = 0; l < n_enc; ++l) {
        a.enc[l] = enc[l];
        if (!enc[l].in_w || !enc[l].out_w || !enc[l].lin1_w || !enc[l].lin2_w) return GOM_ERR_INVALID_ARG;
    }
    for (int l = 0; l < n_dec; ++l) {
        a.dec[l] = dec[l];
        if (!dec[l].in_w || !dec[l].out_w || (dec[l].lin1_w && !dec[l].lin2_w)) return GOM_ERR_INVALID_ARG;
    }
    a.n_enc = n_enc; a.n_dec = n_dec; a.d = d; a.heads = heads; a.ffn = ffn;
    a.img_w = img_w; a.img_h = img_h; a.with_iou = with_iou; a.max_center_dist = max_center_dist;
    a.scale = 1.0f / sqrtf((float)(d / heads));
    const long wide = 3L * d > ffn ? 3L * d : ffn;           // matcher_rt.cpp's carve-up of the workspace
    float* p = workspace;
    a.src = p;    p += (long)N * d;
    a.mem_a = p;  p += (long)N * d;
    a.mem_b = p;  p += (long)N * d;
    a.mem_c = p;  p += (long)N * d;
    a.big = p;    p += (long)N * wide;
    a.att = p;    p += (long)N * d;
    a.tgt_a = p;  p += (long)n_k * d;
    a.tgt_b = p;  p += (long)n_k * d;
    a.qbuf = p;   p += (long)n_k * d;
    a.hid = p;    p += (long)n_k * wide;
    a.logits = p;
    a.traj = traj; a.sync = sync; a.status = status;
    if (desc_host) {
        if (!desc_dev || desc_words <= 0) return GOM_ERR_INVALID_ARG;
        a.desc_src = (const unsigned*)desc_host; a.desc_dst = (unsigned*)desc_dev; a.desc_words = desc_words;
    }
    hipLaunchKernelGGL(match_fused_kernel, dim3((unsigned)g_grid), dim3(THREADS), sizeof(float) * (size_t)N, (hipStream_t)stream, a);
    return gom_launch_status();
}
