// Detection post-processing on device (gfx950): GoMatching.detection() + box/NMS + the association
// head's foreground filter, with fixed nq-padded outputs and a per-frame count so no host sync is
// needed inside a batch of frames.
//   gom_lstmatcher.py:579-629 (scores, rescoring max, threshold, px scaling, char argmax)
//   gom_lstmatcher.py:310-332 (boxes = min/max of boundary points, torchvision-style greedy NMS)
//   lstmatcher.py:271-282     (objectness > asso_thresh_test)
#include "common.h"

namespace {

constexpr int MAXQ = 1024;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// one wave per row: index of the first maximum
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ld, int V, long rows,
                                                          int* __restrict__ out) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float best = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int j = lane; j < V; j += 64) {
        const float v = x[r * ld + j];
        if (v > best) { best = v; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[r] = bi;
}

__global__ __launch_bounds__(256) void detect_post_kernel(
    const float* __restrict__ cls, int ld_cls, const float* __restrict__ recls, int ld_recls,
    const float* __restrict__ ctrl, const float* __restrict__ bd, const int* __restrict__ recs_in, int nq, int P,
    float img_h, float img_w, float det_thr, float nms_thr, float asso_thr, int* __restrict__ count,
    int* __restrict__ keep_idx, float* __restrict__ scores_out, float* __restrict__ boxes_out,
    float* __restrict__ ctrl_out, float* __restrict__ bd_out, long long* __restrict__ recs_out) {
    __shared__ float s_score[MAXQ];
    __shared__ float s_box[MAXQ][4];
    __shared__ int s_order[MAXQ];
    __shared__ unsigned char s_sel[MAXQ], s_dead[MAXQ];
    __shared__ int s_kept[MAXQ];
    __shared__ int s_n, s_nkeep;
    const int b = blockIdx.x, tid = threadIdx.x;

    // 1. scores + selection + pixel-space boxes
    for (int q = tid; q < nq; q += 256) {
        const long base = ((long)b * nq + q) * P;
        float sum = 0.f;
        for (int p = 0; p < P; ++p) sum += cls[(base + p) * ld_cls];
        float sc = sigmoidf_(sum / (float)P);
        if (recls) {
            float rs = 0.f;
            for (int p = 0; p < P; ++p) rs += recls[(base + p) * ld_recls];
            const float re = sigmoidf_(rs / (float)P);
            sc = (sc > re) ? sc : re;
        }
        s_score[q] = sc;
        s_sel[q] = sc > det_thr;
        float x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
        for (int p = 0; p < P; ++p) {
            const float* pt = bd + (base + p) * 4;
            const float ax = pt[0] * img_w, ay = pt[1] * img_h, bx = pt[2] * img_w, by = pt[3] * img_h;
            x0 = fminf(x0, fminf(ax, bx)); x1 = fmaxf(x1, fmaxf(ax, bx));
            y0 = fminf(y0, fminf(ay, by)); y1 = fmaxf(y1, fmaxf(ay, by));
        }
        s_box[q][0] = x0; s_box[q][1] = y0; s_box[q][2] = x1; s_box[q][3] = y1;
    }
    if (tid == 0) { s_n = 0; s_nkeep = 0; }
    __syncthreads();

    // 2. stable descending rank among the selected queries
    for (int q = tid; q < nq; q += 256) {
        if (!s_sel[q]) continue;
        const float sq = s_score[q];
        int rank = 0;
        for (int j = 0; j < nq; ++j)
            if (s_sel[j] && (s_score[j] > sq || (s_score[j] == sq && j < q))) ++rank;
        s_order[rank] = q;
        atomicAdd(&s_n, 1);
    }
    __syncthreads();
    const int n = s_n;
    for (int i = tid; i < n; i += 256) s_dead[i] = 0;
    __syncthreads();

    // 3. greedy NMS in score order (IoU > thr suppresses)
    for (int i = 0; i < n; ++i) {
        if (!s_dead[i]) {                                   // uniform: written before the last barrier
            const int qi = s_order[i];
            const float ix0 = s_box[qi][0], iy0 = s_box[qi][1], ix1 = s_box[qi][2], iy1 = s_box[qi][3];
            const float iarea = (ix1 - ix0) * (iy1 - iy0);
            for (int j = i + 1 + tid; j < n; j += 256) {
                if (s_dead[j]) continue;
                const int qj = s_order[j];
                const float w = fmaxf(0.f, fminf(ix1, s_box[qj][2]) - fmaxf(ix0, s_box[qj][0]));
                const float h = fmaxf(0.f, fminf(iy1, s_box[qj][3]) - fmaxf(iy0, s_box[qj][1]));
                const float inter = w * h;
                const float area = (s_box[qj][2] - s_box[qj][0]) * (s_box[qj][3] - s_box[qj][1]);
                if (inter / (iarea + area - inter) > nms_thr) s_dead[j] = 1;
            }
        }
        __syncthreads();
    }

    // 4. ordered compaction + association-head foreground filter
    if (tid == 0) {
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (!s_dead[i] && s_score[s_order[i]] > asso_thr) s_kept[m++] = s_order[i];
        s_nkeep = m;
        count[b] = m;
    }
    __syncthreads();
    const int m = s_nkeep;

    // 5. gather the kept instances (padded slots are left untouched)
    for (int t = tid; t < m; t += 256) {
        const int q = s_kept[t];
        const long o = (long)b * nq + t;
        keep_idx[o] = b * nq + q;                            // row into the [B*nq, ...] detector tensors
        scores_out[o] = s_score[q];
        boxes_out[o * 4 + 0] = s_box[q][0]; boxes_out[o * 4 + 1] = s_box[q][1];
        boxes_out[o * 4 + 2] = s_box[q][2]; boxes_out[o * 4 + 3] = s_box[q][3];
    }
    for (int u = tid; u < m * P; u += 256) {
        const int t = u / P, p = u % P;
        const int q = s_kept[t];
        const long src = ((long)b * nq + q) * P + p, dst = ((long)b * nq + t) * P + p;
        ctrl_out[dst * 2] = ctrl[src * 2] * img_w;
        ctrl_out[dst * 2 + 1] = ctrl[src * 2 + 1] * img_h;
        bd_out[dst * 4] = bd[src * 4] * img_w;
        bd_out[dst * 4 + 1] = bd[src * 4 + 1] * img_h;
        bd_out[dst * 4 + 2] = bd[src * 4 + 2] * img_w;
        bd_out[dst * 4 + 3] = bd[src * 4 + 3] * img_h;
        recs_out[dst] = (long long)recs_in[src];
    }
}

}  // namespace

extern "C" int gom_argmax_rows_f32(const float* x, int ld, int V, long rows, int* out, void* stream) {
    GOM_CHECK_ARG(x && out && V > 0 && ld >= V && rows >= 0);
    if (rows == 0) return GOM_OK;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, V,
                       rows, out);
    return gom_launch_status();
}

extern "C" int gom_detect_post(const float* cls_logits, int ld_cls, const float* rescoring_logits, int ld_rescoring,
                               const float* ctrl_points, const float* bd_points, const int* recs, int B,
                               int num_queries, int num_points, float img_h, float img_w, float det_thresh,
                               float nms_thresh, float asso_thresh, int* count, int* keep_idx, float* scores,
                               float* boxes, float* ctrl_out, float* bd_out, long long* recs_out, void* stream) {
    GOM_CHECK_ARG(cls_logits && ctrl_points && bd_points && recs && count && keep_idx && scores && boxes && ctrl_out &&
                  bd_out && recs_out);
    GOM_CHECK_ARG(B > 0 && num_queries > 0 && num_queries <= MAXQ && num_points > 0 && ld_cls >= 1);
    hipLaunchKernelGGL(detect_post_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, cls_logits, ld_cls,
                       rescoring_logits, ld_rescoring, ctrl_points, bd_points, recs, num_queries, num_points, img_h,
                       img_w, det_thresh, nms_thresh, asso_thresh, count, keep_idx, scores, boxes, ctrl_out, bd_out,
                       recs_out);
    return gom_launch_status();
}
