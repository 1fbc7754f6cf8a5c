// Per-frame association records of the multi-GPU exchange (gomatching_amd/dist.py): ONE launch packs a step's
// detections -- re-id rows of the pool, boxes, scores, control points, boundary points, characters -- into the fixed-shape
// fp32 buffer [F, nq + 1, D] that goes through the all-gather; row 0 of a frame carries (count, image height, image width).
// The per-frame Python loop this replaces issued six slice-assign kernels per frame on the tracker stream, which at 8 GPUs is
// the critical path.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pack_records_kernel(const float* __restrict__ pool, int ld_pool, int row_base,
                                                           const int* __restrict__ counts, const float* __restrict__ boxes,
                                                           const float* __restrict__ scores, const float* __restrict__ ctrl,
                                                           const float* __restrict__ bd, const long* __restrict__ recs,
                                                           int nq, int fd, int P, float img_h, float img_w,
                                                           float* __restrict__ out) {
    const int f = blockIdx.y, i = blockIdx.x;                // frame, record row (0 = header)
    const int D = fd + 5 + 7 * P;
    float* dst = out + ((size_t)f * (nq + 1) + i) * D;
    const int n = counts[f];
    if (i == 0) {
        for (int c = threadIdx.x; c < D; c += 256) dst[c] = c == 0 ? (float)n : (c == 1 ? img_h : (c == 2 ? img_w : 0.f));
        return;
    }
    const int k = i - 1;
    if (k >= n) {
        for (int c = threadIdx.x; c < D; c += 256) dst[c] = 0.f;
        return;
    }
    int prev = 0;                                            // pool rows of earlier frames of the step (<= 64 frames)
    for (int g = 0; g < f; ++g) prev += counts[g];
    const float* re = pool + (size_t)(row_base + prev + k) * ld_pool;
    const size_t q = (size_t)f * nq + k;                     // slot in the nq-padded detection arrays
    for (int c = threadIdx.x; c < D; c += 256) {
        float v;
        if (c < fd) v = re[c];
        else if (c < fd + 4) v = boxes[q * 4 + (c - fd)];
        else if (c == fd + 4) v = scores[q];
        else if (c < fd + 5 + 2 * P) v = ctrl[q * 2 * P + (c - fd - 5)];
        else if (c < fd + 5 + 6 * P) v = bd[q * 4 * P + (c - fd - 5 - 2 * P)];
        else v = (float)recs[q * P + (c - fd - 5 - 6 * P)];
        dst[c] = v;
    }
}

}  // namespace

extern "C" int gom_pack_records_f32(const float* pool, int ld_pool, int row_base, const int* counts, const float* boxes,
                                    const float* scores, const float* ctrl, const float* bd, const long* recs, int frames,
                                    int nq, int feature_dim, int num_points, float img_h, float img_w, float* out,
                                    void* stream) {
    GOM_CHECK_ARG(pool && counts && boxes && scores && ctrl && bd && recs && out);
    GOM_CHECK_ARG(frames >= 0 && nq > 0 && feature_dim > 0 && num_points > 0 && ld_pool >= feature_dim && row_base >= 0);
    if (frames == 0) return GOM_OK;
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)(nq + 1), (unsigned)frames), dim3(256), 0, (hipStream_t)stream,
                       pool, ld_pool, row_base, counts, boxes, scores, ctrl, bd, recs, nq, feature_dim, num_points, img_h,
                       img_w, out);
    return gom_launch_status();
}
