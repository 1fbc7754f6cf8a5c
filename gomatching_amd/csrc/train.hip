// Element-wise / row-wise pieces of TRAINING the association head on the device (only `roi_heads` trains in
// the reference, freeze_layers.py:20-37): the backward of ReLU and of a scaled row softmax, the per-frame cross entropy with a
// background logit of `detr_asso_loss` (lstmatcher.py:436-475) with its gradient, and the sigmoid focal loss of `loss_res`
// (lstmatcher.py:237-268) with its gradient.  The contractions (Linear / attention products and their dgrad / wgrad) run on
// the exact-fp32 MFMA GEMM with transposed operands (gomatching_amd/training.py).  Latency-bound, tiny tensors.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void relu_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            float* __restrict__ dx, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// dS[r, j] = scale * P[r, j] * (dP[r, j] - sum_k dP[r, k] P[r, k]),   one wave per row
__global__ __launch_bounds__(256) void softmax_backward_kernel(const float* __restrict__ P, const float* __restrict__ dP,
                                                               float* __restrict__ dS, long rows, int cols, long ld,
                                                               float scale) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = P + r * ld;
    const float* g = dP + r * ld;
    float dot = 0.f;
    for (int j = lane; j < cols; j += 64) dot = fmaf(g[j], p[j], dot);
    dot = wave_sum(dot);
    for (int j = lane; j < cols; j += 64) dS[r * ld + j] = scale * p[j] * (g[j] - dot);
}

// Per (row i, frame t): cross entropy over [logits[i, offs[t]:offs[t+1]] | 0] against gt[i*T + t] (index inside the frame,
// n_t = the background class; < 0 = the pair does not count).  loss[i*T + t] = logsumexp - logit of the target;
// grad != nullptr: dlogits[i, lo + j] = g * (softmax_j - [j == target]) (zero for pairs that do not count).
__global__ __launch_bounds__(256) void asso_ce_kernel(const float* __restrict__ logits, int ld, const int* __restrict__ offs,
                                                      int T, const int* __restrict__ gt, long rows, float* __restrict__ loss,
                                                      const float* __restrict__ gscale, float* __restrict__ dlogits) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows * T) return;
    const int lane = threadIdx.x & 63;
    const long i = w / T;
    const int t = (int)(w % T);
    const int lo = offs[t], hi = offs[t + 1], n = hi - lo;
    const int target = gt[w];
    const float* row = logits + i * ld + lo;
    if (target < 0) {
        if (loss && lane == 0) loss[w] = 0.f;
        if (dlogits)
            for (int j = lane; j < n; j += 64) dlogits[i * ld + lo + j] = 0.f;
        return;
    }
    float mx = 0.f;                                          // the appended background logit
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, row[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += expf(row[j] - mx);
    sum = wave_sum(sum) + expf(0.f - mx);
    if (loss && lane == 0) loss[w] = (mx + logf(sum)) - (target < n ? row[target] : 0.f);
    if (dlogits) {
        const float g = gscale[0];
        for (int j = lane; j < n; j += 64)
            dlogits[i * ld + lo + j] = g * (expf(row[j] - mx) / sum - (j == target ? 1.f : 0.f));
    }
}

// Sigmoid focal loss per element (adet sigmoid_focal_loss as used by loss_res): t in {0, 1};
//   p_t = sigmoid((2t - 1) x), loss = alpha_t (1 - p_t)^gamma (-log p_t), dloss/dx = (2t - 1) alpha_t (1 - p_t)^gamma (gamma p_t log p_t - (1 - p_t))
__global__ __launch_bounds__(256) void sigmoid_focal_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            float alpha, float gamma, long n, float* __restrict__ loss,
                                                            float* __restrict__ dx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float tt = t[i], sgn = 2.f * tt - 1.f, s = sgn * x[i];
    // -log sigmoid(s) = softplus(-s), stable for both signs (binary_cross_entropy_with_logits)
    const float nlp = fmaxf(-s, 0.f) + log1pf(expf(-fabsf(s)));
    const float pt = expf(-nlp);
    const float a = alpha >= 0.f ? (alpha * tt + (1.f - alpha) * (1.f - tt)) : 1.f;
    const float mod = powf(1.f - pt, gamma);
    if (loss) loss[i] = a * mod * nlp;
    if (dx) dx[i] = sgn * a * mod * (gamma * pt * (-nlp) - (1.f - pt));
}

}  // namespace

extern "C" int gom_relu_backward_f32(const float* dy, const float* y, float* dx, long n, void* stream) {
    GOM_CHECK_ARG(n >= 0 && (n == 0 || (dy && y && dx)));
    if (n == 0) return GOM_OK;
    hipLaunchKernelGGL(relu_backward_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    return gom_launch_status();
}

extern "C" int gom_softmax_rows_backward_f32(const float* P, const float* dP, float* dS, long rows, int cols, long ld,
                                             float scale, void* stream) {
    GOM_CHECK_ARG(rows >= 0 && cols >= 0 && ld >= cols && (rows == 0 || cols == 0 || (P && dP && dS)));
    if (rows == 0 || cols == 0) return GOM_OK;
    hipLaunchKernelGGL(softmax_backward_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, P, dP, dS,
                       rows, cols, ld, scale);
    return gom_launch_status();
}

extern "C" int gom_asso_ce_f32(const float* logits, int ld, const int* frame_offsets, int num_frames, const int* gt, long rows,
                               float* loss, const float* grad_scale, float* dlogits, void* stream) {
    GOM_CHECK_ARG(rows >= 0 && num_frames > 0 && frame_offsets && (rows == 0 || (logits && gt)));
    GOM_CHECK_ARG((loss || dlogits) && (!dlogits || grad_scale));
    if (rows == 0) return GOM_OK;
    hipLaunchKernelGGL(asso_ce_kernel, dim3((unsigned)cdiv(rows * num_frames, 4)), dim3(256), 0, (hipStream_t)stream, logits,
                       ld, frame_offsets, num_frames, gt, rows, loss, grad_scale, dlogits);
    return gom_launch_status();
}

extern "C" int gom_sigmoid_focal_f32(const float* x, const float* target, float alpha, float gamma, long n, float* loss,
                                     float* dx, void* stream) {
    GOM_CHECK_ARG(n >= 0 && (n == 0 || (x && target && (loss || dx))));
    if (n == 0) return GOM_OK;
    hipLaunchKernelGGL(sigmoid_focal_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, target, alpha,
                       gamma, n, loss, dx);
    return gom_launch_status();
}
