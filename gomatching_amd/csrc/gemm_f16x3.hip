// fp32-class GEMM / implicit-GEMM convolution on the fp16 matrix cores ("f16x3" split emulation).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  fp32 in, fp32 out
//
// Each fp32 operand is split into TWO fp16 planes,  x0 = fp16(x), x1 = fp16(x - x0)  (x - x0 is exact in fp32), i.e.
// 22 significand bits, and the product is rebuilt from three plane products
//   a.b ~= a0b0 + a0b1 + a1b0            (the dropped a1b1 is <= 2^-22 |a||b|)
// accumulated in fp32 by v_mfma_f32_32x32x16_f16: HALF the MFMA passes, 2/3 of the LDS traffic and half the split
// VALU of the bf16x6 kernel (gemm_bf16x6.hip), for a relative error of ~3 * 2^-22 = 7e-7 per product against ~2e-7
// there -- both far inside what an fp32 GEMM's own accumulation order moves a K >= 64 dot product by (sqrt(K) * 2^-24).
// fp16 has a 5-bit exponent, which the split has to respect:
//   * weights are constants: each row is scaled by an exact power of two into fp16's upper normal range before the
//     split (gom_split_f16x2; the inverse scale is applied in the epilogue through `wscale`), so both planes are
//     normal numbers and the 2^-22 bound holds whatever the weight magnitude;
//   * activations are split as they are: magnitudes up to 65504 are representable; below ~0.25 the second plane
//     becomes subnormal and the element keeps an ABSOLUTE accuracy of 2^-25 = 3e-8 (fp32's spacing at 0.5) instead
//     of a relative one.  An activation beyond fp16's range would become Inf: every tile checks its results and raises
//     a device flag (`flag`), which the host turns into an error at the step's sync point -- never a silent wrong
//     result.  Badly scaled data (the 12-decade rows of tests/test_ops_gpu.py) belongs on the bf16x6 kernel.
// Same tiling, staging, epilogue, XCD-aware tile order, implicit-im2col addressing and split-K form as gemm_bf16x6.hip,
// but THREE workgroups per CU (two planes -> 40 KB of LDS, 138-147 VGPRs with one k-tile of A in flight).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int ROW_BYTES = 80;                    // 32 bf16 + 16 B pad: 5 sixteen-byte slots (odd) per row

struct Args {
    const float* A;
    const unsigned short* Wp;                    // [2][N][ldw] fp16 planes (rows pre-scaled by 2^e_n)
    const float* wscale;                         // [N] 2^-e_n, applied to the accumulator before scale/shift
    int* flag;                                   // device word, set non-zero when a result is not finite
    long w_plane_stride;                         // elements between planes
    float* C;
    const float* scale;
    const float* shift;
    const float* R;
    const int* a_rows;
    int M, N, K;
    int lda, ldw, ldc, ldr;
    int relu;
    int r_cols;                                  // residual applies to columns < r_cols
    int r_period;                                // > 0: row m reads residual row m % r_period (a table shared by the frames)
    int H, Wd, cin_log2, OH, OW, stride, pad;
    // split-K (few output tiles, long K): blockIdx.y owns k-tiles [y*kt_per_split, ...) and stores raw partial sums
    int kt_per_split;
    float* partial;                              // [splits][M][N] or nullptr
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// two floats -> packed fp16 pair (v_cvt_pk_f16_f32, round to nearest even) and the exact residuals
__device__ __forceinline__ void split2(float x, float y, unsigned int& q0, unsigned int& q1) { gom_split2_f16(x, y, q0, q1); }
// split 4 floats into 2 planes of 4 fp16 (8 bytes each)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& p0, u32x2& p1) {
    unsigned int a0, a1, b0, b1;
    split2(v[0], v[1], a0, a1);
    split2(v[2], v[3], b0, b1);
    p0 = u32x2{a0, b0};
    p1 = u32x2{a1, b1};
}

template <int BM, int BN, int KH, int KW, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_f16x3_kernel(const Args p) {
    constexpr int WM = BM / 2, WN = BN / 2;                  // 2 x 2 waves
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_UNITS = BM * 8 / 256;                    // float4 units per thread per k-tile (8 per row)
    constexpr int W_UNITS = BN * 4 / 256;                    // 16-byte units per thread per plane (4 per row)
    constexpr bool CONV = KH > 0;
    constexpr int A_PLANE = BM * ROW_BYTES, W_PLANE = BN * ROW_BYTES;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                                // [2][BM][80 B]
    unsigned char* Ws = smem + 2 * A_PLANE;                  // [2][BN][80 B]

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

    // ---- buffer descriptors: 32-bit byte offsets, out-of-range lanes read zeros (no select instructions) ----
    constexpr unsigned RANGE = 0x80000000u, INVALID = 0xC0000000u;       // offsets stay OOB after adding < 1 GiB
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (int)RANGE, 0x00020000);

    // A: unit u = tid + i*256 -> row u>>3, k-quad u&7.  a_off = byte offset of (row, k = kq*4) or of the tap origin
    const int kq = tid & 7;
    unsigned a_off[A_UNITS];
    int a_ih0[A_UNITS], a_iw0[A_UNITS];
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
        const int row = (tid >> 3) + i * 32;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        if (CONV) {
            const int ow = mm % p.OW;
            const int t = mm / p.OW;
            const int oh = t % p.OH;
            const int b = t / p.OH;
            a_ih0[i] = ok ? oh * p.stride - p.pad : -(1 << 28);            // invalid row: every tap fails the bounds test
            a_iw0[i] = ow * p.stride - p.pad;
            a_off[i] = (unsigned)((((b * p.H + oh * p.stride - p.pad) * p.Wd + a_iw0[i]) << p.cin_log2) * 4);
        } else {
            const int src = p.a_rows ? p.a_rows[mm] : mm;
            a_off[i] = ok ? (unsigned)(src * p.lda + kq * 4) * 4u : INVALID;
            a_ih0[i] = a_iw0[i] = 0;
        }
    }
    // W: unit u = tid + i*256 -> row u>>2, 16-byte chunk u&3 (8 bf16); byte offsets into plane 0
    const int wq = tid & 3;
    unsigned w_off[W_UNITS];
#pragma unroll
    for (int i = 0; i < W_UNITS; ++i) {
        const int n = n0 + (tid >> 2) + i * 64;
        w_off[i] = n < p.N ? (unsigned)(n * p.ldw + wq * 8) * 2u : INVALID;
    }
    const unsigned w_plane_bytes = (unsigned)(p.w_plane_stride * 2);

    f32x4 a_even[A_UNITS], a_odd[A_UNITS];                 // two k-tiles of A in flight (HBM latency > one MFMA phase)
    u32x4 w_reg[2][W_UNITS];

    const int nk_all = (p.K + BK - 1) / BK;
    const int ktb = p.partial ? (int)blockIdx.y * p.kt_per_split : 0;   // first k-tile of this workgroup
    const int nk = p.partial ? max(0, min(nk_all - ktb, p.kt_per_split)) : nk_all;

    auto load_A = [&](int kt_rel, f32x4 (&a_reg)[A_UNITS]) {
        const int kt = kt_rel + ktb;
        unsigned koff = (unsigned)(kt * BK) * 4u;            // plain GEMM: columns kt*BK..
        const bool k_ok = kt * BK + kq * 4 < p.K;            // K tail reads nothing (weights are zero there anyway)
        int kh = 0, kw = 0;
        if (CONV) {
            const int k = kt * BK + kq * 4;
            const int c = k & ((1 << p.cin_log2) - 1);
            const int khw = k >> p.cin_log2;
            kh = khw / KW;
            kw = khw - kh * KW;
            koff = (unsigned)((((kh * p.Wd + kw) << p.cin_log2) + c) * 4);
            if (k >= p.K) kh = 1 << 28;                      // K tail: force out of range
        }
#pragma unroll
        for (int i = 0; i < A_UNITS; ++i) {
            unsigned off = a_off[i] + koff;
            if (!CONV && !k_ok) off = INVALID;
            if (CONV) {
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                if (!(((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.Wd))) off = INVALID;
            }
            a_reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)off, 0, 0));
        }
    };
    auto load_W = [&](int kt_rel) {
        const unsigned koff = (unsigned)((kt_rel + ktb) * BK) * 2u;      // planes are zero-padded to ldw (multiple of 32)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < W_UNITS; ++i)
                w_reg[pl][i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)(w_off[i] + koff + pl * w_plane_bytes), 0, 0);
    };
    auto store_tile = [&](const f32x4 (&a_reg)[A_UNITS]) {  // split A in registers, then A planes + W planes -> LDS
#pragma unroll
        for (int i = 0; i < A_UNITS; ++i) {
            const int row = (tid >> 3) + i * 32;
            u32x2 p0, p1;
            split4(a_reg[i], p0, p1);
            unsigned char* d = As + row * ROW_BYTES + kq * 8;
            *reinterpret_cast<u32x2*>(d) = p0;
            *reinterpret_cast<u32x2*>(d + A_PLANE) = p1;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
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

    const int fr = lane & 31, fh = lane >> 5;
    const unsigned char* a_base = As + (wr * WM + fr) * ROW_BYTES + fh * 16;
    const unsigned char* w_base = Ws + (wc * WN + fr) * ROW_BYTES + fh * 16;

    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8 af[2][MT];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[pl][i] = *reinterpret_cast<const half8*>(a_base + pl * A_PLANE + i * 32 * ROW_BYTES + ks * 32);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const half8 b0 = *reinterpret_cast<const half8*>(w_base + j * 32 * ROW_BYTES + ks * 32);
                const half8 b1 = *reinterpret_cast<const half8*>(w_base + W_PLANE + j * 32 * ROW_BYTES + ks * 32);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x16 c = acc[i][j];                    // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], b0, c, 0, 0, 0);
                    acc[i][j] = c;
                }
            }
        }
    };

    if constexpr (OCC >= 3) {
        // three workgroups per CU (168 VGPRs): one k-tile of A in flight; the other two workgroups cover the latency
        if (nk > 0) {
            load_A(0, a_even);
            load_W(0);
            store_tile(a_even);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                load_W(kt + 1);
                load_A(kt + 1, a_even);
            }
            compute();
            __syncthreads();
            if (kt + 1 < nk) {
                store_tile(a_even);
                __syncthreads();
            }
        }
    } else {
        if (nk > 0) {
            load_A(0, a_even);
            load_W(0);
            if (nk > 1) load_A(1, a_odd);
            store_tile(a_even);
        }
        __syncthreads();

        for (int kt = 0; kt < nk; kt += 2) {
            // tile kt is in LDS, a_odd holds tile kt+1 (issued a whole iteration ago)
            if (kt + 1 < nk) load_W(kt + 1);
            if (kt + 2 < nk) load_A(kt + 2, a_even);
            compute();
            __syncthreads();
            if (kt + 1 >= nk) break;
            store_tile(a_odd);
            __syncthreads();
            // tile kt+1 is in LDS, a_even holds tile kt+2
            if (kt + 2 < nk) load_W(kt + 2);
            if (kt + 3 < nk) load_A(kt + 3, a_odd);
            compute();
            __syncthreads();
            if (kt + 2 < nk) {
                store_tile(a_even);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: y = acc*scale + shift (+ residual) (ReLU) --------------------------------------
    const float relu_lo = p.relu == 1 ? 0.f : -INFINITY;     // relu: 0 none, 1 ReLU, 2 GELU
    int bad = 0;
    if (p.partial) {                                         // split-K: raw sums, the reducer applies the epilogue
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
    const bool vec_ok = ((p.N | p.ldc) & 3) == 0 && (!p.R || (p.ldr & 3) == 0);
    if (vec_ok) {
        // stage each 32-row slab of the wave's patch through (now free) LDS so that global traffic is whole
        // 16-byte pieces of contiguous rows: 4x fewer store/load instructions than the per-register pattern
        constexpr int ES = WN + 4;                           // padded row (floats)
        float* stage = reinterpret_cast<float*>(smem) + wave * (32 * ES);
        constexpr int C4 = WN / 4;                           // float4 per row
        constexpr int RPI = 64 / C4;                         // rows covered per wave-instruction
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[((r & 3) + 8 * (r >> 2) + 4 * fh) * ES + j * 32 + fr] = acc[i][j][r];
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the slab is wave-private
            const int c4 = (lane % C4) * 4;
            const int n = n0 + wc * WN + c4;
            const bool n_ok = n < p.N;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (n_ok && p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (n_ok && p.wscale) sc = sc * *reinterpret_cast<const f32x4*>(p.wscale + n);   // exact: a power of two
            if (n_ok && p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            const bool use_r = p.R && n < p.r_cols;
            f32x4 rv[32 / RPI];
#pragma unroll
            for (int t = 0; t < 32 / RPI; ++t) {
                const int m = m0 + wr * WM + i * 32 + t * RPI + lane / C4;
                rv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (use_r && n_ok && m < p.M)
                    rv[t] = *reinterpret_cast<const f32x4*>(p.R + (size_t)(p.r_period > 0 ? m % p.r_period : m) * p.ldr + n);
            }
#pragma unroll
            for (int t = 0; t < 32 / RPI; ++t) {
                const int row = t * RPI + lane / C4;
                const int m = m0 + wr * WM + i * 32 + row;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * ES + c4);
                v = v * sc + sh + rv[t];
                // range check on the PRE-activation value: fmaxf(NaN, 0) = 0 would hide the Inf - Inf of an operand beyond fp16
                const bool bad_v = !(fabsf(v[0]) <= 3.4e38f) | !(fabsf(v[1]) <= 3.4e38f) | !(fabsf(v[2]) <= 3.4e38f) |
                                   !(fabsf(v[3]) <= 3.4e38f);
                if (p.relu == 2) {                           // exact GELU (Swin MLP, swin_transformer.py:36-38)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
                } else {
                    v[0] = fmaxf(v[0], relu_lo); v[1] = fmaxf(v[1], relu_lo);
                    v[2] = fmaxf(v[2], relu_lo); v[3] = fmaxf(v[3], relu_lo);
                }
                if (n_ok && m < p.M) {
                    bad |= bad_v;
                    *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);              // reads done before the next slab overwrites
        }
        if (bad && p.flag) atomicOr(p.flag, 1);              // Inf / NaN: an operand left fp16's range (or came in bad)
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wc * WN + j * 32 + fr;
        const bool n_ok = n < p.N;
        const float sc = ((n_ok && p.scale) ? p.scale[n] : 1.f) * ((n_ok && p.wscale) ? p.wscale[n] : 1.f);
        const float sh = (n_ok && p.shift) ? p.shift[n] : 0.f;
        const bool use_r = p.R && n < p.r_cols;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n_ok && m < p.M) {
                    float v = acc[i][j][r] * sc + sh;
                    if (use_r) v += p.R[(size_t)(p.r_period > 0 ? m % p.r_period : m) * p.ldr + n];
                    bad |= !(fabsf(v) <= 3.4e38f);             // before the activation (see above)
                    v = p.relu == 2 ? 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)) : fmaxf(v, relu_lo);
                    p.C[(size_t)m * p.ldc + n] = v;
                }
            }
        }
    }
    if (bad && p.flag) atomicOr(p.flag, 1);
}

// sums the split-K partials in split order (deterministic) and applies the epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const Args p, int splits) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)p.M * p.N) return;
    const int m = (int)(i / p.N), n = (int)(i % p.N);
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += p.partial[(size_t)s * p.M * p.N + i];
    v = v * ((p.scale ? p.scale[n] : 1.f) * (p.wscale ? p.wscale[n] : 1.f)) + (p.shift ? p.shift[n] : 0.f);
    if (p.R && n < p.r_cols) v += p.R[(size_t)(p.r_period > 0 ? m % p.r_period : m) * p.ldr + n];
    if (!(fabsf(v) <= 3.4e38f) && p.flag) atomicOr(p.flag, 1);   // before the activation: fmaxf(NaN, 0) = 0
    if (p.relu == 1) v = fmaxf(v, 0.f);
    if (p.relu == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    p.C[(size_t)m * p.ldc + n] = v;
}

template <int BM, int BN, int KH, int KW>
int launch(const Args& a, hipStream_t s, int splits = 1) {
    const long tiles = (long)cdiv(a.M, BM) * cdiv(a.N, BN);
    if (tiles <= 0) return GOM_OK;
    constexpr int lds_loop = 2 * (BM + BN) * ROW_BYTES, lds_epi = 4 * 32 * (BN / 2 + 4) * 4;   // k-loop planes | epilogue slabs
    const int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    // three workgroups per CU (40 KB LDS, <= 168 VGPRs, one k-tile of A in flight) measured 9 % faster end to end than
    // two with a two-deep A prefetch (251 VGPRs): 224-299 vs 187-265 TFLOP/s on the encoder shapes
    auto kern = gemm_f16x3_kernel<BM, BN, KH, KW, 3>;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(256), lds, s, a);
    if (a.partial)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv((long)a.M * a.N, 256)), dim3(256), 0, s, a, splits);
    return gom_launch_status();
}

template <int KH, int KW>
int dispatch(const Args& a, hipStream_t s, int splits = 1) {
    if (a.N <= 64) return launch<128, 64, KH, KW>(a, s, splits);
    return launch<128, 128, KH, KW>(a, s, splits);
}

// fp32 [N, ldw_in] -> two fp16 planes [2][N][Kpad] (zero padded in K) of the row scaled by 2^e_n, e_n chosen so that the
// row's largest magnitude lands in [2^13, 2^14); inv_scale[n] = 2^-e_n.  One workgroup per row.
__global__ __launch_bounds__(256) void split_rows_f16_kernel(const float* __restrict__ W, int ldw, int N, int K,
                                                             unsigned short* __restrict__ out, int Kpad,
                                                             float* __restrict__ inv_scale) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float* row = W + (size_t)n * ldw;
    float mx = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(row[k]));
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (mx > 0.f && mx <= 3.4e38f) {
        int ex;
        frexpf(mx, &ex);                                     // mx = f * 2^ex, f in [0.5, 1)
        e = 14 - ex;                                         // mx * 2^e in [2^13, 2^14)
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    const float sc = ldexpf(1.f, e);
    if (threadIdx.x == 0) inv_scale[n] = ldexpf(1.f, -e);
    const long plane = (long)N * Kpad;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        const float x = k < K ? row[k] * sc : 0.f;           // exact: power-of-two scaling
        const _Float16 h0 = (_Float16)x;
        const _Float16 h1 = (_Float16)(x - (float)h0);
        out[(size_t)n * Kpad + k] = __builtin_bit_cast(unsigned short, h0);
        out[plane + (size_t)n * Kpad + k] = __builtin_bit_cast(unsigned short, h1);
    }
}

}  // namespace

extern "C" int gom_split_f16x2(const float* W, int ldw, int N, int K, void* planes_out, int Kpad, float* inv_scale,
                               void* stream) {
    GOM_CHECK_ARG(W && planes_out && inv_scale && N > 0 && K > 0 && ldw >= K && Kpad >= K && (Kpad % 32) == 0);
    hipLaunchKernelGGL(split_rows_f16_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                       (unsigned short*)planes_out, Kpad, inv_scale);
    return gom_launch_status();
}

extern "C" int gom_gemm_f32_f16x3(const float* A, const int* a_rows, int lda, const void* Wplanes, long w_plane_stride,
                                  int ldw, const float* wscale, const float* scale, const float* shift, const float* R,
                                  int ldr, int r_cols, int relu, float* C, int ldc, int M, int N, int K, int* flag,
                                  void* stream) {
    return gom_gemm_f32_f16x3_rp(A, a_rows, lda, Wplanes, w_plane_stride, ldw, wscale, scale, shift, R, ldr, r_cols, 0, relu, C,
                                 ldc, M, N, K, flag, stream);
}

extern "C" int gom_gemm_f32_f16x3_rp(const float* A, const int* a_rows, int lda, const void* Wplanes, long w_plane_stride,
                                     int ldw, const float* wscale, const float* scale, const float* shift, const float* R,
                                     int ldr, int r_cols, int r_period, int relu, float* C, int ldc, int M, int N, int K,
                                     int* flag, void* stream) {
    GOM_CHECK_ARG(A && Wplanes && C && r_period >= 0);
    GOM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 4) == 0);
    GOM_CHECK_ARG((lda % 4) == 0 && lda >= K && (ldw % 32) == 0 && ldw >= K && ldc >= N && (w_plane_stride % 8) == 0);
    GOM_CHECK_ARG(!R || (r_cols > 0 && r_cols <= N && ldr >= r_cols));
    GOM_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)Wplanes % 16) == 0 && ((uintptr_t)C % 16) == 0);
    GOM_CHECK_ARG((long)M * lda < (1L << 29) || a_rows);        // 32-bit byte offsets with a 2 GiB range check
    if (M == 0) return GOM_OK;
    Args a{};
    a.A = A; a.r_cols = r_cols; a.r_period = R ? r_period : 0; a.Wp = (const unsigned short*)Wplanes; a.w_plane_stride = w_plane_stride; a.C = C;
    a.wscale = wscale; a.flag = flag;
    a.scale = scale; a.shift = shift; a.R = R; a.a_rows = a_rows;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.relu = relu;
    return dispatch<0, 0>(a, (hipStream_t)stream);
}

extern "C" int gom_conv2d_nhwc_f32_f16x3(const float* X, const void* Wplanes, long w_plane_stride, int ldw,
                                         const float* wscale, const float* scale, const float* shift, const float* R,
                                         int relu, float* Y, int B, int H, int Wd, int Cin, int Cout, int KH, int KW,
                                         int stride, int pad, void* workspace, long workspace_bytes, int splits,
                                         int* flag, void* stream) {
    GOM_CHECK_ARG(X && Wplanes && Y);
    GOM_CHECK_ARG(B > 0 && H > 0 && Wd > 0 && Cin >= 4 && Cout > 0 && stride > 0 && pad >= 0);
    GOM_CHECK_ARG((Cin & (Cin - 1)) == 0);
    GOM_CHECK_ARG(KH == KW && (KH == 1 || KH == 3 || KH == 7));
    GOM_CHECK_ARG((long)B * H * Wd * Cin < (1L << 29));
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    GOM_CHECK_ARG(OH > 0 && OW > 0);
    int lg = 0;
    while ((1 << lg) < Cin) ++lg;
    Args a{};
    a.r_cols = Cout;
    a.A = X; a.Wp = (const unsigned short*)Wplanes; a.w_plane_stride = w_plane_stride; a.C = Y;
    a.wscale = wscale; a.flag = flag;
    a.scale = scale; a.shift = shift; a.R = R; a.relu = relu;
    a.M = B * OH * OW; a.N = Cout; a.K = KH * KW * Cin;
    a.lda = Cin; a.ldw = ldw; a.ldc = Cout; a.ldr = Cout;
    a.H = H; a.Wd = Wd; a.cin_log2 = lg; a.OH = OH; a.OW = OW; a.stride = stride; a.pad = pad;
    GOM_CHECK_ARG(ldw >= a.K && (ldw % 32) == 0 && (long)a.M * Cout < (1L << 31));
    hipStream_t s = (hipStream_t)stream;
    if (splits > 1) {
        GOM_CHECK_ARG(workspace && workspace_bytes >= (long)sizeof(float) * splits * a.M * a.N);
        a.partial = (float*)workspace;
        a.kt_per_split = cdiv(cdiv(a.K, BK), splits);
    } else {
        splits = 1;
    }
    if (KH == 1 && stride == 1 && pad == 0) {
        a.H = a.Wd = a.OH = a.OW = 0;
        return dispatch<0, 0>(a, s, splits);
    }
    if (KH == 1) return dispatch<1, 1>(a, s, splits);
    if (KH == 3) return dispatch<3, 3>(a, s, splits);
    return dispatch<7, 7>(a, s, splits);
}
