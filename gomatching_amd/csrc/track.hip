// Association-score kernels of the LST-Matcher tracker (gfx950).  Latency-bound, tiny tensors: the
// float arithmetic of run_short_term_match / run_long_term_match runs here, the integer id
// bookkeeping and the assignment problem stay on the host exactly as in the reference.
//   lstmatcher.py:373-381            _activate_asso (per-frame softmax with a zero background logit)
//   gom_lstmatcher.py:429-445,510-547 trajectory score, last-box IoU, time decay, centre-distance gate
#include "common.h"
#include "tracker_tasks.h"

namespace {

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows,
                                                          float* __restrict__ out, int n, int d4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)n * d4) return;
    const int r = (int)(i / d4), c = (int)(i % d4);
    *reinterpret_cast<f32x4*>(out + i * 4) = *reinterpret_cast<const f32x4*>(src + ((size_t)rows[r] * d4 + c) * 4);
}

// A match with hoisted projections (matcher_rt.cpp): one launch gathers the window's embeddings [N, d], their precomputed encoder
// in-projections [N, 3d] and the current frame's precomputed decoder query projections [n_k, d] (rows lo..hi-1 of the window).
__global__ __launch_bounds__(256) void gather_match_kernel(const float* __restrict__ pool, int ld_pool,
                                                           const float* __restrict__ proj, int ld_proj,
                                                           const int* __restrict__ rows, int N, int lo, int n_k, int d4,
                                                           float* __restrict__ src, float* __restrict__ qkv,
                                                           float* __restrict__ qdec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    gom_tasks::gather_match_item(pool, ld_pool, proj, ld_proj, rows, N, lo, n_k, d4, src, qkv, qdec, i);
}

// one wave per (query row, frame segment)
__global__ __launch_bounds__(256) void asso_activate_kernel(const float* __restrict__ logits, int ld,
                                                            const int* __restrict__ offs, int T, int n_k,
                                                            float* __restrict__ out, int ld_out) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (long)n_k * T) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(w / T), t = (int)(w % T);
    const int lo = offs[t], hi = offs[t + 1];
    const float* row = logits + (size_t)i * ld;
    float mx = 0.f;                                          // the appended background logit
    for (int j = lo + lane; j < hi; j += 64) mx = fmaxf(mx, row[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lo + lane; j < hi; j += 64) sum += expf(row[j] - mx);
    sum = wave_sum(sum) + expf(0.f - mx);
    for (int j = lo + lane; j < hi; j += 64) out[(size_t)i * ld_out + j] = expf(row[j] - mx) / sum;
}

// meta layout (int32): nonk[Np] | col_of[Np] | last_idx[M] | k_inds[n_k]; the score itself: gom_tasks::track_score_one (tracker_tasks.h)
using gom_tasks::track_score_one;

__global__ __launch_bounds__(256) void track_score_kernel(const float* __restrict__ act, int ld,
                                                          const int* __restrict__ meta, const float* __restrict__ decay,
                                                          const float* __restrict__ boxes, float img_w, float img_h,
                                                          int n_k, int Np, int M, int with_iou, float max_center_dist,
                                                          float* __restrict__ traj) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_k * M) return;
    const int i = idx / M, m = idx % M;
    traj[idx] = track_score_one(act + (size_t)i * ld, meta, decay, boxes, img_w, img_h, i, m, Np, M, with_iou, max_center_dist);
}

// asso_activate + track_score of ONE match in one launch (the two kernels above back to back cost two dependent launches per
// frame; beside the detector every tracker launch also delays the detector's next kernel).  One workgroup per
// current detection i: the waves run `asso_activate_kernel`'s code over the frame segments into an LDS row, then the threads
// call `track_score_one` over the tracks on that row -- the same arithmetic, value for value.
__global__ __launch_bounds__(256) void asso_score_kernel(const float* __restrict__ logits, int ld,
                                                         const int* __restrict__ offs, int T, const int* __restrict__ meta,
                                                         const float* __restrict__ decay, const float* __restrict__ boxes,
                                                         float img_w, float img_h, int n_k, int Np, int M, int with_iou,
                                                         float max_center_dist, float* __restrict__ traj) {
    extern __shared__ float act[];                           // [N] activations of detection i
    gom_tasks::asso_score_block(logits, ld, offs, T, meta, decay, boxes, img_w, img_h, Np, M, with_iou, max_center_dist, traj,
                                (int)blockIdx.x, act, (int)threadIdx.x, 256);
}

// Short-term matching, all (previous, current) frame pairs of a clip in one launch (gom_lstmatcher.py:405-445 with the
// id-independent reading of roi_heads.short_term_scores): one wave per current-frame detection i of pair p computes
//   logits l_j = tgt_i . memory_j over the PREVIOUS frame's rows j (ATTWeightHead, 0 layers),
//   a_j = softmax over {l_j} U {0} (lstmatcher.py:373-381; the current frame's own block never feeds a track score),
//   S[i, j] = max(a_j, IoU(box_i, box_j)) (with_iou) -- the trajectory score of a track seen once.
// pair descriptors [device] int32 [P][6] = (first memory row, n_prev, n_cur, first tgt row, first box row, S offset);
// row_pair [total cur rows] = pair of each tgt row.
constexpr int ST_CHUNKS = 5;                                 // previous-frame detections per pair <= 320

__global__ __launch_bounds__(256) void short_term_pairs_kernel(const float* __restrict__ tgt, const float* __restrict__ mem,
                                                               int d, const int* __restrict__ pairs,
                                                               const int* __restrict__ row_pair,
                                                               const float* __restrict__ boxes, float img_w, float img_h,
                                                               int with_iou, int total_rows, float* __restrict__ S) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= total_rows) return;
    const int lane = threadIdx.x & 63;
    const int* p = pairs + 6 * row_pair[w];
    const int m0 = p[0], n_prev = p[1], t0 = p[3], b0 = p[4];
    const int i = w - t0;
    const float* a = tgt + (size_t)w * d;
    float* out = S + p[5] + (size_t)i * n_prev;
    // logits stay in registers: lane l keeps columns l, l+64, ... (n_prev <= 64 * ST_CHUNKS, checked by the host)
    float lg[ST_CHUNKS];
    float mx = 0.f;
#pragma unroll
    for (int c = 0; c < ST_CHUNKS; ++c) {
        lg[c] = -INFINITY;
        const int jend = min(n_prev, (c + 1) * 64);
        for (int j = c * 64; j < jend; ++j) {
            const float* b = mem + (size_t)(m0 + j) * d;
            float acc = 0.f;
            for (int k = lane * 4; k < d; k += 256) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(a + k);
                const f32x4 y = *reinterpret_cast<const f32x4*>(b + k);
                acc = fmaf(x[0], y[0], acc);
                acc = fmaf(x[1], y[1], acc);
                acc = fmaf(x[2], y[2], acc);
                acc = fmaf(x[3], y[3], acc);
            }
            acc = wave_sum(acc);
            mx = fmaxf(mx, acc);
            if (lane == (j & 63)) lg[c] = acc;
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < ST_CHUNKS; ++c)
        if (c * 64 + lane < n_prev) sum += expf(lg[c] - mx);
    sum = wave_sum(sum) + expf(0.f - mx);
    const float* kb = boxes + (size_t)(b0 + n_prev + i) * 4;
    const float kx0 = kb[0] / img_w, ky0 = kb[1] / img_h, kx1 = kb[2] / img_w, ky1 = kb[3] / img_h;
#pragma unroll
    for (int c = 0; c < ST_CHUNKS; ++c) {
        const int j = c * 64 + lane;
        if (j >= n_prev) continue;
        float s = expf(lg[c] - mx) / sum;
        if (with_iou) {
            const float* lb = boxes + (size_t)(b0 + j) * 4;
            const float lx0 = lb[0] / img_w, ly0 = lb[1] / img_h, lx1 = lb[2] / img_w, ly1 = lb[3] / img_h;
            const float ww = fmaxf(fminf(kx1, lx1) - fmaxf(kx0, lx0), 0.f);
            const float hh = fmaxf(fminf(ky1, ly1) - fmaxf(ky0, ly0), 0.f);
            const float inter = ww * hh;
            const float a1 = (kx1 - kx0) * (ky1 - ky0), a2 = (lx1 - lx0) * (ly1 - ly0);
            const float iou = inter > 0.f ? inter / (a1 + a2 - inter) : 0.f;
            s = fmaxf(s, iou);
        }
        out[j] = s;
    }
}

}  // namespace

extern "C" int gom_gather_rows_f32(const float* src, const int* rows, float* out, int n, int dim, void* stream) {
    GOM_CHECK_ARG(src && rows && out && n >= 0 && dim > 0 && (dim % 4) == 0);
    if (n == 0) return GOM_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv((long)n * dim / 4, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, rows, out, n, dim / 4);
    return gom_launch_status();
}

extern "C" int gom_gather_match_f32(const float* pool, int ld_pool, const float* proj, int ld_proj, const int* rows, int N,
                                    int lo, int n_k, int dim, float* src, float* qkv, float* qdec, void* stream) {
    GOM_CHECK_ARG(pool && proj && rows && src && qkv && qdec && N > 0 && lo >= 0 && n_k >= 0 && lo + n_k <= N);
    GOM_CHECK_ARG(dim > 0 && (dim % 4) == 0 && ld_pool >= dim && ld_proj >= 4 * dim && (ld_pool % 4) == 0 && (ld_proj % 4) == 0);
    const long quads = ((long)N * 4 + n_k) * (dim / 4);
    hipLaunchKernelGGL(gather_match_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, (hipStream_t)stream, pool, ld_pool,
                       proj, ld_proj, rows, N, lo, n_k, dim / 4, src, qkv, qdec);
    return gom_launch_status();
}

extern "C" int gom_asso_activate_f32(const float* logits, int ld, const int* frame_offsets, int num_frames, int n_k,
                                     float* out, int ld_out, void* stream) {
    GOM_CHECK_ARG(logits && frame_offsets && out && num_frames > 0 && n_k >= 0);
    if (n_k == 0) return GOM_OK;
    hipLaunchKernelGGL(asso_activate_kernel, dim3((unsigned)cdiv((long)n_k * num_frames, 4)), dim3(256), 0,
                       (hipStream_t)stream, logits, ld, frame_offsets, num_frames, n_k, out, ld_out);
    return gom_launch_status();
}

extern "C" int gom_track_score_f32(const float* act, int ld, const int* meta, const float* decay, const float* boxes,
                                   float img_w, float img_h, int n_k, int Np, int M, int with_iou,
                                   float max_center_dist, float* traj, void* stream) {
    GOM_CHECK_ARG(act && meta && boxes && traj && n_k >= 0 && Np >= 0 && M >= 0);
    if (n_k == 0 || M == 0) return GOM_OK;
    hipLaunchKernelGGL(track_score_kernel, dim3((unsigned)cdiv((long)n_k * M, 256)), dim3(256), 0, (hipStream_t)stream,
                       act, ld, meta, decay, boxes, img_w, img_h, n_k, Np, M, with_iou, max_center_dist, traj);
    return gom_launch_status();
}

/* gom_asso_activate_f32 followed by gom_track_score_f32 as one launch (same values); logits [n_k, ld] over the N = Np + n_k
 * selected detections of the window.  N is bounded by the LDS row: N <= 16 384. */
extern "C" int gom_asso_score_f32(const float* logits, int ld, const int* frame_offsets, int num_frames, const int* meta,
                                  const float* decay, const float* boxes, float img_w, float img_h, int n_k, int Np, int M,
                                  int with_iou, float max_center_dist, float* traj, void* stream) {
    GOM_CHECK_ARG(logits && frame_offsets && meta && boxes && traj && num_frames > 0 && n_k >= 0 && Np >= 0 && M >= 0);
    GOM_CHECK_ARG(ld >= Np + n_k && Np + n_k <= 16384);
    if (n_k == 0 || M == 0) return GOM_OK;
    hipLaunchKernelGGL(asso_score_kernel, dim3((unsigned)n_k), dim3(256), sizeof(float) * (size_t)(Np + n_k), (hipStream_t)stream,
                       logits, ld, frame_offsets, num_frames, meta, decay, boxes, img_w, img_h, n_k, Np, M, with_iou,
                       max_center_dist, traj);
    return gom_launch_status();
}

extern "C" int gom_short_term_pairs_f32(const float* tgt, const float* memory, int d, const int* pairs,
                                        const int* row_pair, const float* boxes, float img_w, float img_h, int with_iou,
                                        int total_cur_rows, int max_prev, float* S, void* stream) {
    GOM_CHECK_ARG(tgt && memory && pairs && row_pair && boxes && S && d > 0 && (d % 4) == 0 && total_cur_rows >= 0);
    GOM_CHECK_ARG(max_prev >= 0 && max_prev <= 64 * ST_CHUNKS);
    if (total_cur_rows == 0) return GOM_OK;
    hipLaunchKernelGGL(short_term_pairs_kernel, dim3((unsigned)cdiv(total_cur_rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       tgt, memory, d, pairs, row_pair, boxes, img_w, img_h, with_iou, total_cur_rows, S);
    return gom_launch_status();
}
