// fp32 GEMM / implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[M,N] = epilogue( (A [+ A2])[M,K] . W[N,K]^T )
//
// Covers every dense contraction on the GoMatching path: R-50 convolutions with
// FrozenBN folded into the epilogue (A2), input_proj convs (A4), all nn.Linear layers of the
// deformable encoder/decoder (A6, A9), the heads (A8, A10, A11), FCHead4Query (A13) and the
// matcher transformers / association dot product (A14, A15).
//
// Design (gfx950):
//   * exact-fp32 MFMA: the result is a k-ordered fmaf chain, which is what the 1e-3 end-to-end /
//     identical-top-k requirement needs (no bf16 anywhere).
//   * block tile BM x BN x 16, 4 waves (one per SIMD), each wave a (WM x WN) patch of 32x32 MFMA
//     tiles; operands staged global -> registers -> LDS (double-buffered, one barrier per k-tile);
//     LDS rows padded to 20 floats so the ds_read_b128 fragment reads are bank-conflict free.
//   * each lane reads 4 consecutive k per row (one ds_read_b128) and feeds 4 MFMAs: MFMA j of a
//     k-step covers k = {j, 4+j}; A and W use the same slot->k map so the product is unchanged.
//   * implicit im2col: activations are NHWC, weights OHWI, so a 16-wide k-tile is a contiguous
//     channel run of one (kh,kw) tap (Cin is a power of two; the RGB stem is padded to 4).
//   * XCD-aware block order: consecutive tiles (which share the A panel / the whole W) are dealt to
//     the same XCD so their re-reads hit that XCD's L2.
#include "common.h"

namespace {

constexpr int BK = 16;
constexpr int LDS_STRIDE = 20;  // floats per staged row (16 + 4 pad)

struct GemmArgs {
    const float* A;
    const float* A2;
    const float* W;
    float* C;
    const float* scale;
    const float* shift;
    const float* R;
    const int* a_rows;
    int M, N, K;
    int lda, ldw, ldc, ldr;
    int relu;
    // convolution geometry (KH > 0)
    int H, Wd, cin_log2, OH, OW, stride, pad;
    // split-K (skinny problems): blockIdx.y owns k-tiles [y*kt_per_split, ...) and stores raw partial sums
    int kt_per_split;
    float* partial;                                          // [splits][M][N] or nullptr
};

template <int BM, int BN, int WM, int WN, int KH, int KW>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_UNITS = BM * 4 / 256;                    // float4 loads per thread per k-tile
    constexpr int W_UNITS = (BN * 4 + 255) / 256;
    constexpr bool CONV = KH > 0;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                        // [2][BM][LDS_STRIDE]
    float* Ws = smem + 2 * BM * LDS_STRIDE;                  // [2][BN][LDS_STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;

    // ---- XCD-aware tile order (bijective for any grid size) -------------------------------
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread load descriptors --------------------------------------------------------
    const int kq = tid & 3;                                  // which float4 of the 16-wide k-tile
    int a_off[A_UNITS];                                      // element offset of the row (or tap origin)
    int a_ih0[A_UNITS], a_iw0[A_UNITS];
    bool a_ok[A_UNITS];
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
        const int row = (tid >> 2) + i * 64;
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if (CONV) {
            const int ow = mm % p.OW;
            const int t = mm / p.OW;
            const int oh = t % p.OH;
            const int b = t / p.OH;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
            a_off[i] = ((b * p.H + a_ih0[i]) * p.Wd + a_iw0[i]) << p.cin_log2;
        } else {
            const int src = p.a_rows ? p.a_rows[mm] : mm;
            a_off[i] = src * p.lda;
            a_ih0[i] = a_iw0[i] = 0;
        }
    }
    int w_off[W_UNITS];
    bool w_ok[W_UNITS];
#pragma unroll
    for (int i = 0; i < W_UNITS; ++i) {
        const int row = (tid >> 2) + i * 64;
        const int n = n0 + row;
        w_ok[i] = (row < BN) && (n < p.N);
        w_off[i] = (w_ok[i] ? n : 0) * p.ldw;
    }

    f32x4 a_reg[A_UNITS], w_reg[W_UNITS];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + kq * 4;
        const bool k_ok = k < p.K;
        int tap_off = k;
        int kh = 0, kw = 0;
        if (CONV) {
            const int c = k & ((1 << p.cin_log2) - 1);
            const int khw = k >> p.cin_log2;
            kh = khw / KW;
            kw = khw - kh * KW;
            tap_off = ((kh * p.Wd + kw) << p.cin_log2) + c;
        }
#pragma unroll
        for (int i = 0; i < A_UNITS; ++i) {
            bool ok = a_ok[i] && k_ok;
            if (CONV) {
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.Wd);
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                v = *reinterpret_cast<const f32x4*>(p.A + a_off[i] + tap_off);
                if (p.A2) v += *reinterpret_cast<const f32x4*>(p.A2 + a_off[i] + tap_off);
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_UNITS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w_ok[i] && k_ok) v = *reinterpret_cast<const f32x4*>(p.W + w_off[i] + k);
            w_reg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_UNITS; ++i) {
            const int row = (tid >> 2) + i * 64;
            *reinterpret_cast<f32x4*>(As + (buf * BM + row) * LDS_STRIDE + kq * 4) = a_reg[i];
        }
#pragma unroll
        for (int i = 0; i < W_UNITS; ++i) {
            const int row = (tid >> 2) + i * 64;
            if (row < BN) *reinterpret_cast<f32x4*>(Ws + (buf * BN + row) * LDS_STRIDE + kq * 4) = w_reg[i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = p.partial ? blockIdx.y * p.kt_per_split : 0;
    const int nk = p.partial ? min(nk_all, kt0 + p.kt_per_split) : nk_all;
    const int fr = lane & 31, fh = lane >> 5;

    load_tile(kt0);
    store_tile(kt0 & 1);
    __syncthreads();

    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);                  // global loads fly under the MFMAs
        const float* a_base = As + (buf * BM + wr * WM + fr) * LDS_STRIDE + fh * 4;
        const float* w_base = Ws + (buf * BN + wc * WN + fr) * LDS_STRIDE + fh * 4;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDS_STRIDE + ks * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(w_base + j * 32 * LDS_STRIDE + ks * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    if (p.partial) {                                         // split-K: raw partial sums, epilogue in the reducer
        float* dst = p.partial + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wc * WN + j * 32 + fr;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (n < p.N && m < p.M) dst[(size_t)m * p.N + n] = acc[i][j][r];
                }
        }
        return;
    }
    // ---- epilogue: y = acc*scale + shift (+ residual) (ReLU) ---------------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wc * WN + j * 32 + fr;
        const bool n_ok = n < p.N;
        const float sc = (n_ok && p.scale) ? p.scale[n] : 1.f;
        const float sh = (n_ok && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n_ok && m < p.M) {
                    float v = acc[i][j][r];
                    if (p.scale) v = v * sc + sh; else v = v + sh;
                    if (p.R) v += p.R[(size_t)m * p.ldr + n];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.C[(size_t)m * p.ldc + n] = v;
                }
            }
        }
    }
}

// sums the split-K partials in split order (deterministic) and applies the epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p, int splits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)p.M * p.N) return;
    const int m = (int)(i / p.N), n = (int)(i % p.N);
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += p.partial[(size_t)s * p.M * p.N + i];
    const float sh = p.shift ? p.shift[n] : 0.f;
    if (p.scale) v = v * p.scale[n] + sh; else v = v + sh;
    if (p.R) v += p.R[(size_t)m * p.ldr + n];
    if (p.relu) v = fmaxf(v, 0.f);
    p.C[(size_t)m * p.ldc + n] = v;
}

template <int BM, int BN, int WM, int WN, int KH, int KW>
int launch(const GemmArgs& a, hipStream_t s) {
    const long tiles = (long)cdiv(a.M, BM) * cdiv(a.N, BN);
    if (tiles <= 0) return GOM_OK;
    const size_t lds = 2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, KH, KW>), dim3((unsigned)tiles), dim3(256), lds, s, a);
    return gom_launch_status();
}

template <int KH, int KW>
int dispatch_tile(const GemmArgs& a, hipStream_t s) {
    if (a.N <= 32) return launch<256, 32, 64, 32, KH, KW>(a, s);
    if (a.N <= 64) return launch<256, 64, 64, 64, KH, KW>(a, s);
    return launch<128, 128, 64, 64, KH, KW>(a, s);
}

}  // namespace

extern "C" int gom_gemm_f32(const float* A, const float* A2, const int* a_rows, int lda, const float* W, int ldw,
                            const float* scale, const float* shift, const float* R, int ldr, int relu, float* C,
                            int ldc, int M, int N, int K, void* stream) {
    GOM_CHECK_ARG(A && W && C);
    GOM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 4) == 0);
    GOM_CHECK_ARG((lda % 4) == 0 && (ldw % 4) == 0 && lda >= K && ldw >= K && ldc >= N);
    GOM_CHECK_ARG(!R || ldr >= N);
    GOM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && (!A2 || ((uintptr_t)A2 % 16) == 0));
    GOM_CHECK_ARG((long)M * lda < (1L << 31) || a_rows);
    if (M == 0) return GOM_OK;
    GemmArgs a{};
    a.A = A; a.A2 = A2; a.W = W; a.C = C; a.scale = scale; a.shift = shift; a.R = R; a.a_rows = a_rows;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.relu = relu;
    return dispatch_tile<0, 0>(a, (hipStream_t)stream);
}

static int splitk_splits(int K) {                           // ~8 k-tiles of 16 per workgroup, at most 32 slices
    const int nk = cdiv(K, BK);
    int s = nk / 8;
    return s < 1 ? 1 : (s > 32 ? 32 : s);
}

extern "C" long gom_gemm_splitk_workspace_bytes(int M, int N, int K) {
    return (long)sizeof(float) * splitk_splits(K) * M * N;
}

// Skinny problems (M <= 128: tracker / re-id head, weight-read bound): K is split over blockIdx.y so that
// N/64 x 8 workgroups stream the weights in parallel instead of N/128 workgroups looping the whole K.
extern "C" int gom_gemm_f32_splitk(const float* A, const int* a_rows, int lda, const float* W, int ldw,
                                   const float* scale, const float* shift, const float* R, int ldr, int relu,
                                   float* C, int ldc, int M, int N, int K, void* workspace, long workspace_bytes,
                                   void* stream) {
    GOM_CHECK_ARG(A && W && C && workspace);
    GOM_CHECK_ARG(M > 0 && N > 0 && K > 0 && (K % 4) == 0 && (long)M * lda < (1L << 31));
    GOM_CHECK_ARG((lda % 4) == 0 && (ldw % 4) == 0 && lda >= K && ldw >= K && ldc >= N && (!R || ldr >= N));
    GOM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    GOM_CHECK_ARG(workspace_bytes >= gom_gemm_splitk_workspace_bytes(M, N, K));
    const int nk = cdiv(K, BK);
    const int splits = splitk_splits(K);
    GemmArgs a{};
    a.A = A; a.W = W; a.C = C; a.scale = scale; a.shift = shift; a.R = R; a.a_rows = a_rows;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.relu = relu;
    a.kt_per_split = cdiv(nk, splits);
    a.partial = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(cdiv(M, 256) * cdiv(N, 64)), (unsigned)splits);
    const size_t lds = 2 * (256 + 64) * LDS_STRIDE * sizeof(float);
    hipLaunchKernelGGL((gemm_f32_kernel<256, 64, 64, 64, 0, 0>), grid, dim3(256), lds, s, a);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv((long)M * N, 256)), dim3(256), 0, s, a, splits);
    return gom_launch_status();
}

extern "C" int gom_conv2d_nhwc_f32(const float* X, const float* Wt, const float* scale, const float* shift,
                                   const float* R, int relu, float* Y, int B, int H, int Wd, int Cin, int Cout,
                                   int KH, int KW, int stride, int pad, void* stream) {
    GOM_CHECK_ARG(X && Wt && Y);
    GOM_CHECK_ARG(B > 0 && H > 0 && Wd > 0 && Cin >= 4 && Cout > 0 && stride > 0 && pad >= 0);
    GOM_CHECK_ARG((Cin & (Cin - 1)) == 0);
    GOM_CHECK_ARG(KH == KW && (KH == 1 || KH == 3 || KH == 7));
    GOM_CHECK_ARG((long)B * H * Wd * Cin < (1L << 31));
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    GOM_CHECK_ARG(OH > 0 && OW > 0);
    int lg = 0;
    while ((1 << lg) < Cin) ++lg;
    GemmArgs a{};
    a.A = X; a.W = Wt; a.C = Y; a.scale = scale; a.shift = shift; a.R = R; a.relu = relu;
    a.M = B * OH * OW; a.N = Cout; a.K = KH * KW * Cin;
    a.lda = Cin; a.ldw = a.K; a.ldc = Cout; a.ldr = Cout;
    a.H = H; a.Wd = Wd; a.cin_log2 = lg; a.OH = OH; a.OW = OW; a.stride = stride; a.pad = pad;
    GOM_CHECK_ARG((long)a.M * Cout < (1L << 31));
    hipStream_t s = (hipStream_t)stream;
    if (KH == 1 && stride == 1 && pad == 0) {                // pointwise conv on channels-last = plain GEMM over pixels
        a.H = a.Wd = a.OH = a.OW = 0;
        return dispatch_tile<0, 0>(a, s);
    }
    if (KH == 1) return dispatch_tile<1, 1>(a, s);
    if (KH == 3) return dispatch_tile<3, 3>(a, s);
    return dispatch_tile<7, 7>(a, s);
}
