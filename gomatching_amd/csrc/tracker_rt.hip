// Native runtime of the per-frame id recurrence (gom_lstmatcher.py:366-564): the loop of
// `GoMatching.track_frames` -- short-term assignment from the precomputed score matrices, long-term match for frames that
// keep unmatched detections (selection, descriptors, the device chain of matcher_rt.cpp, host LSA, thresholds, id
// allocation) -- for ALL frames of a call behind one FFI crossing.  The arithmetic on the device is untouched
// (gom_match_scores_f32); what moves here is the host bookkeeping, ~0.25 ms of interpreter time per frame, which --
// replicated over the 8N frames of an N-GPU step -- is what bounds multi-GPU scaling once the detector runs as a graph
// (tools/tracker_profile.py).  Integer logic follows the numpy calls of meta_arch.py one for one (np.unique = sorted
// distinct, np.searchsorted = lower bound, stable argsort, np.isin, np.maximum.at), so the ids are identical.
//
// Transfers: descriptors go up through a pinned staging buffer and a copy kernel, scores come down through a copy kernel
// into pinned memory behind one stream synchronisation (no pageable staging, no DMA-to-kernel hand-over).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

struct Tracker {
    int test_len, not_mult_thresh, with_iou, n_enc, n_dec, d, heads, ffn;
    float overlap_thresh, max_center_dist;
    int use_decay;
    gom_matcher_layer enc[4], dec[4];
    // device / pinned buffers (grown on demand)
    int* desc_dev = nullptr;     long desc_cap = 0;        // 32-bit words
    int* desc_pin = nullptr;     long desc_pin_cap = 0;
    float* ws_dev = nullptr;     long ws_cap = 0;          // floats
    float* traj_pin = nullptr;   long traj_pin_cap = 0;
    const float* proj = nullptr; int ld_proj = 0;          // hoisted projections for the next run (gom_tracker_set_projections)
};

template <typename T>
int grow_dev(T*& p, long& cap, long need) {
    if (need <= cap) return GOM_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const long want = need + need / 2 + 1024;
    hipError_t e = hipMalloc((void**)&p, sizeof(T) * want);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    cap = want;
    return GOM_OK;
}
template <typename T>
int grow_pin(T*& p, long& cap, long need) {
    if (need <= cap) return GOM_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const long want = need + need / 2 + 1024;
    hipError_t e = hipHostMalloc((void**)&p, sizeof(T) * want, hipHostMallocDefault);
    if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
    cap = want;
    return GOM_OK;
}

// LSA on -traj + thresholds (gom_lstmatcher.py:447-453 / 549-555; GoMatching._assign)
void assign(const Tracker& t, const float* traj, int n_k, const std::vector<long>& uniq, const std::vector<long>& ids_nonk,
            std::vector<long>& out) {
    const int M = (int)uniq.size();
    out.assign(n_k, -1);
    if (n_k == 0 || M == 0) return;
    std::vector<double> cost((size_t)n_k * M);
    for (size_t i = 0; i < cost.size(); ++i) cost[i] = -(double)traj[i];
    const int n = std::min(n_k, M);
    std::vector<long> ri(std::max(n, 1)), ci(std::max(n, 1));
    const int rc = gom_linear_sum_assignment(cost.data(), n_k, M, ri.data(), ci.data());
    for (int a = 0; a < rc; ++a) {
        const long i = ri[a], j = ci[a];
        float thresh = t.overlap_thresh;
        if (!t.not_mult_thresh) {
            long cnt = 0;
            for (long v : ids_nonk) cnt += v == uniq[j];
            thresh = (float)((double)t.overlap_thresh * (double)cnt);     // python float product, then np.float32(...)
        }
        if (traj[(size_t)i * M + j] > thresh) out[i] = uniq[j];
    }
}

std::vector<long> sorted_unique(std::vector<long> v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    return v;
}

int g_sizes = -1;                    // GOM_TRACKER_SIZES=1: one stderr line per long-term match (sizes of the problem)
int g_double_check = -1;             // GOM_TRACKER_DOUBLE_CHECK=1: run every long-term chain twice and compare (diagnostic)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" void* gom_tracker_create(int test_len, float overlap_thresh, int not_mult_thresh, int use_decay, int with_iou,
                                    float max_center_dist, const gom_matcher_layer* enc, int n_enc,
                                    const gom_matcher_layer* dec, int n_dec, int d, int heads, int ffn) {
    if (test_len < 1 || n_enc < 0 || n_enc > 4 || n_dec < 0 || n_dec > 4 || (n_enc && !enc) || (n_dec && !dec) || d <= 0) return nullptr;
    Tracker* t = new Tracker();
    t->test_len = test_len; t->overlap_thresh = overlap_thresh; t->not_mult_thresh = not_mult_thresh; t->use_decay = use_decay;
    t->with_iou = with_iou; t->max_center_dist = max_center_dist; t->n_enc = n_enc; t->n_dec = n_dec; t->d = d; t->heads = heads;
    t->ffn = ffn;
    for (int i = 0; i < n_enc; ++i) t->enc[i] = enc[i];
    for (int i = 0; i < n_dec; ++i) t->dec[i] = dec[i];
    return t;
}

/* Hoisted projections for the NEXT gom_tracker_run call (gom_match_scores_proj_f32): device [pool rows, ld_proj >= 4d], valid
 * for every pool row that call's `rows` name.  NULL (the state after every run) = the chain computes them per match. */
extern "C" int gom_tracker_set_projections(void* h, const float* proj_dev, int ld_proj) {
    Tracker* t = (Tracker*)h;
    if (!t || (proj_dev && ld_proj < 4 * t->d)) return GOM_ERR_INVALID_ARG;
    t->proj = proj_dev;
    t->ld_proj = ld_proj;
    return GOM_OK;
}

extern "C" void gom_tracker_destroy(void* h) {
    Tracker* t = (Tracker*)h;
    if (!t) return;
    if (t->desc_dev) (void)hipFree(t->desc_dev);
    if (t->ws_dev) (void)hipFree(t->ws_dev);
    if (t->desc_pin) (void)hipHostFree(t->desc_pin);
    if (t->traj_pin) (void)hipHostFree(t->traj_pin);
    delete t;
}

/* Frames 0..F-1 of the window handed over (carried frames first, then the new ones from index `first_new`):
 *   n[F], boxes [sum n, 4] (px, host), rows [sum n] (pool rows, host), ids [sum n] (host; carried frames' ids in, new
 *   frames' ids out), S: for every t >= max(first_new, 1) with n[t-1] > 0 and n[t] > 0 the matrix [n[t], n[t-1]] of
 *   precompute_short_term at S + s_off[t] (s_off[t] < 0: no matrix).  `first_real` = absolute frame index of frame
 *   `first_new`; frames before index 0 of the window do not exist for the long-term window (the caller passes at least
 *   test_len - 1 carried frames when the video has them).  decay_table[e] = decay_time ** e as numpy computed it.
 *   secs[0] / secs[1] accumulate the short- / long-term seconds. */
static int tracker_run_impl(void* handle, int F, const int* n, const float* boxes, const int* rows, long* ids, int first_new,
                            long first_real, const float* S, const long* s_off, const float* pool_dev, int ld_pool,
                            float img_w, float img_h, const float* frame_wh, const float* decay_table, long* id_count_io,
                            double* secs, void* stream) {
    Tracker* t = (Tracker*)handle;
    if (!t || F <= 0 || !n || !ids || first_new < 0 || first_new >= F || !id_count_io || !s_off) return GOM_ERR_INVALID_ARG;
    std::vector<long> off(F + 1, 0);
    for (int f = 0; f < F; ++f) {
        if (n[f] < 0) return GOM_ERR_INVALID_ARG;
        off[f + 1] = off[f] + n[f];
    }
    if (off[F] > 0 && (!boxes || !rows)) return GOM_ERR_INVALID_ARG;
    long id_count = *id_count_io;
    hipStream_t st = (hipStream_t)stream;
    if (g_double_check < 0) {
        const char* e = getenv("GOM_TRACKER_DOUBLE_CHECK");
        g_double_check = (e && e[0] == '1') ? 1 : 0;
    }
    std::vector<long> track_ids, uniq, ids_nonk, cur;
    for (int f = first_new; f < F; ++f) {
        const long real = first_real + (f - first_new);
        long* ids_f = ids + off[f];
        const int n_cur = n[f];
        if (real == 0) {                                         // frame 0: ids 1..n, id_count = n + 1 (:377-379)
            for (int i = 0; i < n_cur; ++i) ids_f[i] = i + 1;
            id_count = n_cur + 1;
            continue;
        }
        if (f == 0) return GOM_ERR_INVALID_ARG;                  // a later frame needs its predecessor in the window
        const double t0 = now_s();
        // ---- short-term (run_short_term_match with the precomputed S)
        const int n_prev = n[f - 1];
        const long* ids_prev = ids + off[f - 1];
        std::vector<long> prev(ids_prev, ids_prev + n_prev);
        uniq = sorted_unique(prev);
        const int Ms = (int)uniq.size();
        std::vector<float> traj_s((size_t)n_cur * Ms, 0.f);
        if (n_prev > 0 && n_cur > 0) {
            if (s_off[f] < 0 || !S) return GOM_ERR_INVALID_ARG;
            std::vector<int> order(n_prev);
            for (int j = 0; j < n_prev; ++j) order[j] = j;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return prev[a] < prev[b]; });
            if (Ms != n_prev) return GOM_ERR_INVALID_ARG;         // ids are unique within a frame
            const float* Sf = S + s_off[f];
            for (int i = 0; i < n_cur; ++i)
                for (int m = 0; m < Ms; ++m) traj_s[(size_t)i * Ms + m] = Sf[(size_t)i * n_prev + order[m]];
        }
        assign(*t, traj_s.data(), n_cur, uniq, prev, track_ids);
        if (real == 1) {                                         // id_count given: unmatched detections get new ids now
            for (int i = 0; i < n_cur; ++i)
                if (track_ids[i] < 0) track_ids[i] = ++id_count;
            for (int i = 0; i < n_cur; ++i) ids_f[i] = track_ids[i];
            if (secs) secs[0] += now_s() - t0;
            continue;
        }
        for (int i = 0; i < n_cur; ++i) ids_f[i] = track_ids[i];
        cur = sorted_unique(track_ids);
        const double t1 = now_s();
        if (secs) secs[0] += t1 - t0;
        if (!std::binary_search(cur.begin(), cur.end(), -1L)) continue;
        // ---- long-term (run_long_term_match + _match): window of <= test_len frames ending at f
        const long win_len = std::min<long>(t->test_len, real + 1);
        const int w0 = f + 1 - (int)win_len;
        if (w0 < 0) return GOM_ERR_INVALID_ARG;                  // the caller did not hand over enough carried frames
        const int T = (int)win_len, k = T - 1;
        std::vector<int> sel_idx;                                // absolute detection indices (into boxes / rows / ids)
        std::vector<int> f_sel;
        std::vector<int> n_arr(T, 0);
        for (int w = 0; w < T; ++w) {
            const int fr = w0 + w;
            for (int i = 0; i < n[fr]; ++i) {
                const long id = ids[off[fr] + i];
                const bool sel = (w == k) ? (id == -1) : !std::binary_search(cur.begin(), cur.end(), id);
                if (sel) {
                    sel_idx.push_back((int)(off[fr] + i));
                    f_sel.push_back(w);
                    ++n_arr[w];
                }
            }
        }
        const int N = (int)sel_idx.size(), n_k = n_arr[k], Np = N - n_k;
        ids_nonk.clear();
        for (int a = 0; a < N; ++a)
            if (f_sel[a] != k) ids_nonk.push_back(ids[sel_idx[a]]);
        uniq = sorted_unique(ids_nonk);
        const int M = (int)uniq.size();
        std::vector<long> new_ids(n_k, -1);
        if (g_sizes < 0) {
            const char* e = getenv("GOM_TRACKER_SIZES");
            g_sizes = (e && e[0] == '1') ? 1 : 0;
        }
        if (g_sizes) fprintf(stderr, "MATCH frame %ld T %d N %d n_k %d tracks %d\n", real, T, N, n_k, M);
        if (n_k > 0 && M > 0) {
            // descriptors: rows[N] | offs[T+1] | nonk[Np] col_of[Np] last[M] k_inds[n_k] | boxes[4N] | decay[Np]
            const long words = (long)N + (T + 1) + (2L * Np + M + n_k) + 4L * N + (t->use_decay ? Np : 0);
            int rc = grow_pin(t->desc_pin, t->desc_pin_cap, words);
            if (rc == GOM_OK) rc = grow_dev(t->desc_dev, t->desc_cap, words);
            const long nws = gom_match_workspace_floats(N, n_k, t->d, t->ffn);
            if (rc == GOM_OK) rc = grow_dev(t->ws_dev, t->ws_cap, nws);
            if (rc == GOM_OK) rc = grow_pin(t->traj_pin, t->traj_pin_cap, (long)n_k * M);
            if (rc != GOM_OK) return rc;
            int* p = t->desc_pin;
            int* p_rows = p;                   p += N;
            int* p_offs = p;                   p += T + 1;
            int* p_nonk = p;                   p += Np;
            int* p_col = p;                    p += Np;
            int* p_last = p;                   p += M;
            int* p_kinds = p;                  p += n_k;
            float* p_boxes = (float*)p;        p += 4 * N;
            float* p_decay = (float*)p;
            // `frame_wh` (w, h per window frame): the long-term match normalises EVERY frame's boxes by the image size of the
            // window's FIRST frame (gom_lstmatcher.py:471 `Instances(full_instances[0].image_size)` -> lstmatcher.py:478-494) -- the
            // fp32 quotient is formed here as torch forms it, and the kernels run with an image size of 1 x 1
            float mw = img_w, mh = img_h;
            for (int a = 0; a < N; ++a) {
                p_rows[a] = rows[sel_idx[a]];
                std::memcpy(p_boxes + 4 * a, boxes + 4L * sel_idx[a], 4 * sizeof(float));
                if (frame_wh) {
                    const float fw = frame_wh[2 * w0], fh = frame_wh[2 * w0 + 1];
                    p_boxes[4 * a] /= fw; p_boxes[4 * a + 1] /= fh; p_boxes[4 * a + 2] /= fw; p_boxes[4 * a + 3] /= fh;
                }
            }
            if (frame_wh) mw = mh = 1.f;
            p_offs[0] = 0;
            for (int w = 0; w < T; ++w) p_offs[w + 1] = p_offs[w] + n_arr[w];
            for (int m = 0; m < M; ++m) p_last[m] = 0;
            int jn = 0, jk = 0;
            for (int a = 0; a < N; ++a) {
                if (f_sel[a] != k) {
                    p_nonk[jn] = a;
                    const int c = (int)(std::lower_bound(uniq.begin(), uniq.end(), ids[sel_idx[a]]) - uniq.begin());
                    p_col[jn] = c;
                    if (jn >= 1) p_last[c] = std::max(p_last[c], jn);        // np.maximum.at(last, col_of[1:], arange(1, Np))
                    if (t->use_decay) p_decay[jn] = decay_table[T - 2 - f_sel[a]];
                    ++jn;
                } else {
                    p_kinds[jk++] = a;
                }
            }
            const int* d_rows = t->desc_dev;
            const int* d_offs = d_rows + N;
            const int* d_meta = d_offs + (T + 1);
            const float* d_boxes = (const float*)(d_meta + (2L * Np + M + n_k));
            const float* d_decay = t->use_decay ? d_boxes + 4L * N : nullptr;
            const int lo = p_offs[k];
            // the descriptor copy + the chain of 13 launches; its last phase writes the n_k x M trajectory scores STRAIGHT into pinned
            // host memory (device-visible, posted PCIe writes, complete at the stream sync below).  (The same chain as ONE
            // persistent launch with grid barriers was bit-identical and slower -- 324 vs 132 us per match; tools/exp/match_fused/,
            // docs/LAB_NOTES.md round 5.)
            auto run_match = [&]() -> int {
                const int rc_ = gom_copy_words(t->desc_pin, t->desc_dev, words, stream);
                if (rc_ != GOM_OK) return rc_;
                return gom_match_scores_proj_f32(pool_dev, ld_pool, t->proj, t->ld_proj, d_rows, d_offs, d_meta, d_boxes, d_decay, N, T,
                                                 lo, lo + n_k, M, t->enc, t->n_enc, t->dec, t->n_dec, t->d, t->heads, t->ffn, mw, mh,
                                                 t->with_iou, t->max_center_dist, t->ws_dev, nws, t->traj_pin, stream);
            };
            rc = run_match();
            if (rc != GOM_OK) return rc;
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
            if (g_double_check) {                                // diagnostic: the same match again must give the same bits
                std::vector<float> first(t->traj_pin, t->traj_pin + (size_t)n_k * M);
                rc = run_match();
                if (rc != GOM_OK) return rc;
                if (hipStreamSynchronize(st) != hipSuccess) return GOM_ERR_HIP_BASE;
                if (std::memcmp(first.data(), t->traj_pin, sizeof(float) * (size_t)n_k * M) != 0)
                    fprintf(stderr, "TRAJ MISMATCH at frame %ld (N %d, n_k %d, M %d)\n", real, N, n_k, M);
            }
            assign(*t, t->traj_pin, n_k, uniq, ids_nonk, new_ids);
        }
        for (int i = 0; i < n_k; ++i)
            if (new_ids[i] < 0) new_ids[i] = ++id_count;         // unconditional new ids (:557-560)
        int q = 0;
        for (int i = 0; i < n_cur; ++i)
            if (ids_f[i] == -1) ids_f[i] = new_ids[q++];
        if (secs) secs[1] += now_s() - t1;
    }
    *id_count_io = id_count;
    return GOM_OK;
}

extern "C" int gom_tracker_run(void* handle, int F, const int* n, const float* boxes, const int* rows, long* ids, int first_new,
                               long first_real, const float* S, const long* s_off, const float* pool_dev, int ld_pool,
                               float img_w, float img_h, const float* decay_table, long* id_count_io, double* secs,
                               void* stream) {
    const int rc = tracker_run_impl(handle, F, n, boxes, rows, ids, first_new, first_real, S, s_off, pool_dev, ld_pool, img_w,
                                    img_h, nullptr, decay_table, id_count_io, secs, stream);
    if (handle) ((Tracker*)handle)->proj = nullptr;         // projections are valid for one call only
    return rc;
}

extern "C" int gom_tracker_run_wh(void* handle, int F, const int* n, const float* boxes, const int* rows, long* ids, int first_new,
                                  long first_real, const float* S, const long* s_off, const float* pool_dev, int ld_pool,
                                  const float* frame_wh, const float* decay_table, long* id_count_io, double* secs,
                                  void* stream) {
    if (!frame_wh) return GOM_ERR_INVALID_ARG;
    const int rc = tracker_run_impl(handle, F, n, boxes, rows, ids, first_new, first_real, S, s_off, pool_dev, ld_pool, 1.f, 1.f,
                                    frame_wh, decay_table, id_count_io, secs, stream);
    if (handle) ((Tracker*)handle)->proj = nullptr;
    return rc;
}
