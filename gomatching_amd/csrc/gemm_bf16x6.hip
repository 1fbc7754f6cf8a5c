// fp32-accurate GEMM / implicit-GEMM convolution on the bf16 matrix cores ("bf16x6" split emulation).
//
//   C[M,N] = epilogue( (A [+ A2])[M,K] . W[N,K]^T ),  fp32 in, fp32 out
//
// gfx950 has no TF32-class fast path and its fp32 MFMA runs at 1/16 of the bf16 rate
// (MI355X_MICROARCH.md "Matrix cores").  Each fp32 operand is therefore split into three bf16 planes
//   x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (24 mantissa bits)
// and the product is rebuilt from the six plane products of weight <= 2,
//   a.b ~= a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0),
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are
// <= 3 * 2^-24 |a||b| -- the size of one fp32 rounding -- so results match the exact-fp32 kernel
// (gemm_conv.hip) to accumulation-order noise; tests hold both to the same tolerances.  Six bf16 MFMAs
// per 32x32x16 block cost 192 cycles against 512 for eight v_mfma_f32_32x32x2_f32: 2.67x the fp32 matrix rate.
//
// Weights are constants: they are split ONCE into [3][N][Kpad] bf16 planes (gom_split_bf16x3) and stream
// global -> LDS with no VALU work.  Activations are split in registers on their way to LDS.
// Tile 128x128x32, 4 waves (2x2 MFMA tiles each), one LDS buffer + register prefetch (two barriers per
// k-tile, two workgroups per CU so the other workgroup's MFMAs cover this one's split/store phase), LDS rows
// padded to 80 B so ds_read_b128 fragment reads are conflict-free, same XCD-aware tile order, epilogue and
// implicit-im2col addressing as the fp32 kernel.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int ROW_BYTES = 80;                    // 32 bf16 + 16 B pad: 5 sixteen-byte slots (odd) per row

struct Args {
    const float* A;
    const float* A2;
    const unsigned short* Wp;                    // [3][N][ldw] bf16 planes
    long w_plane_stride;                         // elements between planes
    float* C;
    const float* scale;
    const float* shift;
    const float* R;
    const int* a_rows;
    int M, N, K;
    int lda, ldw, ldc, ldr;
    int relu;
    int H, Wd, cin_log2, OH, OW, stride, pad;
};

__device__ __forceinline__ unsigned int f2bf_bits(float x) {      // round-to-nearest-even (hipcc: v_cvt_pk_bf16_f32)
    return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x);
}
__device__ __forceinline__ float bf_bits2f(unsigned int b) { return __uint_as_float(b << 16); }

// split 4 floats into 3 planes of 4 bf16 (8 bytes each)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
    unsigned int b0[4], b1[4], b2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b0[i] = f2bf_bits(v[i]);
        const float r1 = v[i] - bf_bits2f(b0[i]);
        b1[i] = f2bf_bits(r1);
        const float r2 = r1 - bf_bits2f(b1[i]);
        b2[i] = f2bf_bits(r2);
    }
    p0[0] = b0[0] | (b0[1] << 16); p0[1] = b0[2] | (b0[3] << 16);
    p1[0] = b1[0] | (b1[1] << 16); p1[1] = b1[2] | (b1[3] << 16);
    p2[0] = b2[0] | (b2[1] << 16); p2[1] = b2[2] | (b2[3] << 16);
}

template <int BM, int BN, int KH, int KW>
__global__ __launch_bounds__(256, 2) void gemm_bf16x6_kernel(const Args p) {
    constexpr int WM = BM / 2, WN = BN / 2;                  // 2 x 2 waves
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_UNITS = BM * 8 / 256;                    // float4 units per thread per k-tile (8 per row)
    constexpr int W_UNITS = BN * 4 / 256;                    // 16-byte units per thread per plane (4 per row)
    constexpr bool CONV = KH > 0;
    constexpr int A_PLANE = BM * ROW_BYTES, W_PLANE = BN * ROW_BYTES;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                                // [3][BM][80 B]
    unsigned char* Ws = smem + 3 * A_PLANE;                  // [3][BN][80 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- A descriptors: unit u = tid + i*256 -> row u>>3, k-quad u&7 -----------------------------
    const int kq = tid & 7;
    int a_off[A_UNITS], a_ih0[A_UNITS], a_iw0[A_UNITS];
    bool a_ok[A_UNITS];
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
        const int row = (tid >> 3) + i * 32;
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
    // ---- W descriptors: unit u = tid + i*256 -> row u>>2, 16-byte chunk u&3 (8 bf16) ---------------
    const int wq = tid & 3;
    long w_off[W_UNITS];
    bool w_ok[W_UNITS];
#pragma unroll
    for (int i = 0; i < W_UNITS; ++i) {
        const int row = (tid >> 2) + i * 64;
        const int n = n0 + row;
        w_ok[i] = n < p.N;
        w_off[i] = (long)(w_ok[i] ? n : 0) * p.ldw;
    }

    f32x4 a_reg[A_UNITS];
    u32x4 w_reg[3][W_UNITS];

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
        const int kw8 = kt * BK + wq * 8;                    // planes are zero-padded to ldw (multiple of 32)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < W_UNITS; ++i) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (w_ok[i]) v = *reinterpret_cast<const u32x4*>(p.Wp + pl * p.w_plane_stride + w_off[i] + kw8);
                w_reg[pl][i] = v;
            }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < A_UNITS; ++i) {
            const int row = (tid >> 3) + i * 32;
            u32x2 p0, p1, p2;
            split4(a_reg[i], p0, p1, p2);
            unsigned char* d = As + row * ROW_BYTES + kq * 8;
            *reinterpret_cast<u32x2*>(d) = p0;
            *reinterpret_cast<u32x2*>(d + A_PLANE) = p1;
            *reinterpret_cast<u32x2*>(d + 2 * A_PLANE) = p2;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < W_UNITS; ++i) {
                const int row = (tid >> 2) + i * 64;
                *reinterpret_cast<u32x4*>(Ws + pl * W_PLANE + row * ROW_BYTES + wq * 16) = w_reg[pl][i];
            }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned char* a_base = As + (wr * WM + fr) * ROW_BYTES + fh * 16;
    const unsigned char* w_base = Ws + (wc * WN + fr) * ROW_BYTES + fh * 16;

    load_tile(0);
    store_tile();
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);                  // in flight under the MFMAs below
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[3][MT], bf[3][NT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[pl][i] = *reinterpret_cast<const bf16x8*>(a_base + pl * A_PLANE + i * 32 * ROW_BYTES + ks * 32);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[pl][j] = *reinterpret_cast<const bf16x8*>(w_base + pl * W_PLANE + j * 32 * ROW_BYTES + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    f32x16 c = acc[i][j];                    // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        __syncthreads();                                     // every wave is done reading this tile
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

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

template <int BM, int BN, int KH, int KW>
int launch(const Args& a, hipStream_t s) {
    const long tiles = (long)cdiv(a.M, BM) * cdiv(a.N, BN);
    if (tiles <= 0) return GOM_OK;
    const int lds = 3 * (BM + BN) * ROW_BYTES;
    auto kern = gemm_bf16x6_kernel<BM, BN, KH, KW>;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, s, a);
    return gom_launch_status();
}

template <int KH, int KW>
int dispatch(const Args& a, hipStream_t s) {
    if (a.N <= 64) return launch<128, 64, KH, KW>(a, s);
    return launch<128, 128, KH, KW>(a, s);
}

// fp32 [N, ldw_in] -> three bf16 planes [3][N][Kpad] (zero padded in K)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ W, int ldw, int N, int K,
                                                           unsigned short* __restrict__ out, int Kpad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * Kpad) return;
    const int n = (int)(i / Kpad), k = (int)(i % Kpad);
    const float x = k < K ? W[(size_t)n * ldw + k] : 0.f;
    const unsigned int b0 = f2bf_bits(x);
    const float r1 = x - bf_bits2f(b0);
    const unsigned int b1 = f2bf_bits(r1);
    const float r2 = r1 - bf_bits2f(b1);
    const unsigned int b2 = f2bf_bits(r2);
    const long plane = (long)N * Kpad;
    out[i] = (unsigned short)b0;
    out[plane + i] = (unsigned short)b1;
    out[2 * plane + i] = (unsigned short)b2;
}

}  // namespace

extern "C" int gom_split_bf16x3(const float* W, int ldw, int N, int K, void* planes_out, int Kpad, void* stream) {
    GOM_CHECK_ARG(W && planes_out && N > 0 && K > 0 && ldw >= K && Kpad >= K && (Kpad % 32) == 0);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)cdiv((long)N * Kpad, 256)), dim3(256), 0,
                       (hipStream_t)stream, W, ldw, N, K, (unsigned short*)planes_out, Kpad);
    return gom_launch_status();
}

extern "C" int gom_gemm_f32_bf16x6(const float* A, const float* A2, const int* a_rows, int lda, const void* Wplanes,
                                   long w_plane_stride, int ldw, const float* scale, const float* shift,
                                   const float* R, int ldr, int relu, float* C, int ldc, int M, int N, int K,
                                   void* stream) {
    GOM_CHECK_ARG(A && Wplanes && C);
    GOM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 4) == 0);
    GOM_CHECK_ARG((lda % 4) == 0 && lda >= K && (ldw % 32) == 0 && ldw >= K && ldc >= N && (w_plane_stride % 8) == 0);
    GOM_CHECK_ARG(!R || ldr >= N);
    GOM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)Wplanes % 16) == 0 && (!A2 || ((uintptr_t)A2 % 16) == 0));
    GOM_CHECK_ARG((long)M * lda < (1L << 31) || a_rows);
    if (M == 0) return GOM_OK;
    Args a{};
    a.A = A; a.A2 = A2; a.Wp = (const unsigned short*)Wplanes; a.w_plane_stride = w_plane_stride; a.C = C;
    a.scale = scale; a.shift = shift; a.R = R; a.a_rows = a_rows;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.relu = relu;
    return dispatch<0, 0>(a, (hipStream_t)stream);
}

extern "C" int gom_conv2d_nhwc_f32_bf16x6(const float* X, const void* Wplanes, long w_plane_stride, int ldw,
                                          const float* scale, const float* shift, const float* R, int relu, float* Y,
                                          int B, int H, int Wd, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                          void* stream) {
    GOM_CHECK_ARG(X && Wplanes && Y);
    GOM_CHECK_ARG(B > 0 && H > 0 && Wd > 0 && Cin >= 4 && Cout > 0 && stride > 0 && pad >= 0);
    GOM_CHECK_ARG((Cin & (Cin - 1)) == 0);
    GOM_CHECK_ARG(KH == KW && (KH == 1 || KH == 3 || KH == 7));
    GOM_CHECK_ARG((long)B * H * Wd * Cin < (1L << 31));
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    GOM_CHECK_ARG(OH > 0 && OW > 0);
    int lg = 0;
    while ((1 << lg) < Cin) ++lg;
    Args a{};
    a.A = X; a.Wp = (const unsigned short*)Wplanes; a.w_plane_stride = w_plane_stride; a.C = Y;
    a.scale = scale; a.shift = shift; a.R = R; a.relu = relu;
    a.M = B * OH * OW; a.N = Cout; a.K = KH * KW * Cin;
    a.lda = Cin; a.ldw = ldw; a.ldc = Cout; a.ldr = Cout;
    a.H = H; a.Wd = Wd; a.cin_log2 = lg; a.OH = OH; a.OW = OW; a.stride = stride; a.pad = pad;
    GOM_CHECK_ARG(ldw >= a.K && (ldw % 32) == 0 && (long)a.M * Cout < (1L << 31));
    hipStream_t s = (hipStream_t)stream;
    if (KH == 1 && stride == 1 && pad == 0) {
        a.H = a.Wd = a.OH = a.OW = 0;
        return dispatch<0, 0>(a, s);
    }
    if (KH == 1) return dispatch<1, 1>(a, s);
    if (KH == 3) return dispatch<3, 3>(a, s);
    return dispatch<7, 7>(a, s);
}
