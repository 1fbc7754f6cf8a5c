// fp32-accurate GEMM on the bf16 matrix cores whose ACTIVATION operand already lives in HBM as three bf16 planes
// (same split as gemm_bf16x6.hip: x = x0 + x1 + x2), written there by the producing kernel's epilogue
// (LayerNorm / MSDA / GroupNorm / a previous GEMM).  With both operands pre-split the main loop has no VALU
// work and no register staging at all: tiles stream HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`,
// 1 KiB per wave-instruction), fragments come back by ds_read_b128, six MFMAs per 32x32x16 block.
//
//   C[M,N] (fp32) and/or Cp[3][M,N] (bf16 planes) = epilogue( (A0+A1+A2)[M,K] . (W0+W1+W2)[N,K]^T )
//
// Tile 128x128x32, 4 waves (2x2 patches of 64x64), ONE 48 KiB LDS buffer and three workgroups per CU: a
// workgroup's DMA issue + flight time is covered by the other two workgroups' MFMAs (each SIMD holds one wave
// of each).  The LDS image of an operand plane is [128 rows][64 B], lane-linear as LDS-DMA requires; bank
// conflicts of the fragment reads are removed by XOR-ing the 16-byte chunk index with (row>>2)&3 on BOTH the
// per-lane source address and the read address.  Same accumulation order as gemm_bf16x6.hip, so both kernels
// return identical bits for identical inputs.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;
constexpr int ROWB = BK * 2;                     // bytes per tile row of one plane

struct PArgs {
    const unsigned short* Ap;                    // [3][M][lda] bf16
    const unsigned short* Wp;                    // [3][N][ldw] bf16
    long a_plane_stride, w_plane_stride;         // elements between planes
    float* C;                                    // fp32 output or null
    unsigned short* Cp;                          // plane output or null
    long c_plane_stride;
    const float* scale;
    const float* shift;
    const float* R;
    int M, N, K;
    int lda, ldw, ldc, ldcp, ldr;
    int relu, r_cols;
};

__device__ __forceinline__ unsigned int cvt_pk(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float x, float y, unsigned int& q0, unsigned int& q1, unsigned int& q2) {
    q0 = cvt_pk(x, y);
    const float rx = x - __uint_as_float(q0 << 16), ry = y - __uint_as_float(q0 & 0xFFFF0000u);
    q1 = cvt_pk(rx, ry);
    const float sx = rx - __uint_as_float(q1 << 16), sy = ry - __uint_as_float(q1 & 0xFFFF0000u);
    q2 = cvt_pk(sx, sy);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BM, int BN, int WR, int WC, int OCC>
__global__ __launch_bounds__(64 * WR * WC, OCC) void gemm_planes_kernel(const PArgs p) {
    constexpr int NW = WR * WC;                              // waves: WR x WC patches of WM x WN
    constexpr int WM = BM / WR, WN = BN / WC;
    constexpr int MT = WM / 32, NT = WN / 32;
    constexpr int A_PLANE = BM * ROWB, W_PLANE = BN * ROWB;
    constexpr int A_PER = BM / 16 / NW, W_PER = BN / 16 / NW;            // 1 KiB (16-row) LDS-DMA pieces per wave per plane
    static_assert(A_PER * 16 * NW == BM && W_PER * 16 * NW == BN, "pieces split evenly over the waves");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* As = smem;                                // [3][BM][64 B]
    unsigned char* Ws = smem + 3 * A_PLANE;                  // [3][BN][64 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // one descriptor per plane, sized to the plane's valid rows: rows >= M (N) read zeros
    const unsigned a_bytes = (unsigned)p.M * (unsigned)p.lda * 2u, w_bytes = (unsigned)p.N * (unsigned)p.ldw * 2u;
    __amdgpu_buffer_rsrc_t rsA[3], rsW[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        rsA[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Ap + pl * p.a_plane_stride), 0, (int)a_bytes, 0x00020000);
        rsW[pl] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + pl * p.w_plane_stride), 0, (int)w_bytes, 0x00020000);
    }
    // lane l of a piece fills LDS bytes [16 l, 16 l + 16) = row l>>2, physical chunk l&3, which must hold the
    // logical chunk (l&3) ^ ((row>>2)&3) = (l&3) ^ ((l>>4)&3)   (pieces start at multiples of 16 rows)
    const unsigned lane_row = lane >> 2, lane_chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned a_lane = (unsigned)(m0 + lane_row) * (unsigned)p.lda * 2u + lane_chunk * 16u;
    const unsigned w_lane = (unsigned)(n0 + lane_row) * (unsigned)p.ldw * 2u + lane_chunk * 16u;
    const unsigned a_rb = 16u * (unsigned)p.lda * 2u, w_rb = 16u * (unsigned)p.ldw * 2u;   // bytes per 16-row block

    auto stage = [&](int kt) {
        const unsigned koff = (unsigned)kt * ROWB;
#pragma unroll
        for (int t = 0; t < 3 * A_PER; ++t) {               // plane = compile-time, row block = wave-uniform
            const int pl = t / A_PER, rb = wave * A_PER + t % A_PER;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA[pl], (lds_ptr_t)(As + pl * A_PLANE + rb * 1024), 16,
                                                     (int)(a_lane + rb * a_rb + koff), 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 3 * W_PER; ++t) {
            const int pl = t / W_PER, rb = wave * W_PER + t % W_PER;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW[pl], (lds_ptr_t)(Ws + pl * W_PLANE + rb * 1024), 16,
                                                     (int)(w_lane + rb * w_rb + koff), 0, 0, 0);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 2) & 3;                            // read-side swizzle (row offsets are multiples of 32)
    const unsigned char* a_base = As + (wr * WM + fr) * ROWB;
    const unsigned char* w_base = Ws + (wc * WN + fr) * ROWB;

    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = ((ks * 2 + fh) ^ sw) * 16;
            bf16x8 af[3][MT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[pl][i] = *reinterpret_cast<const bf16x8*>(a_base + pl * A_PLANE + i * 32 * ROWB + ch);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 b0 = *reinterpret_cast<const bf16x8*>(w_base + j * 32 * ROWB + ch);
                bf16x8 b1 = *reinterpret_cast<const bf16x8*>(w_base + W_PLANE + j * 32 * ROWB + ch);
                bf16x8 b2 = *reinterpret_cast<const bf16x8*>(w_base + 2 * W_PLANE + j * 32 * ROWB + ch);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x16 c = acc[i][j];                    // smallest terms first (as gemm_bf16x6.hip)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], b2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], b0, c, 0, 0, 0);
                    acc[i][j] = c;
                }
            }
        }
    };

    for (int kt = 0; kt < nk; ++kt) {
        stage(kt);
        __syncthreads();                                     // vmcnt(0) + barrier: the tile has landed for every wave
        compute();
        __syncthreads();                                     // every wave is done reading before the next DMA lands
    }

    // ---- epilogue: y = acc*scale + shift (+ residual) (ReLU); fp32 rows and/or bf16 planes ---------------
    const float relu_lo = p.relu ? 0.f : -INFINITY;
    constexpr int ES = WN + 4;
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * ES);
    constexpr int C4 = WN / 4;
    constexpr int RPI = 64 / C4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stg[((r & 3) + 8 * (r >> 2) + 4 * fh) * ES + j * 32 + fr] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): the slab is wave-private
        const int c4 = (lane % C4) * 4;
        const int n = n0 + wc * WN + c4;
        const bool n_ok = n < p.N;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (n_ok && p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        if (n_ok && p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
        const bool use_r = p.R && n < p.r_cols;
        f32x4 rv[32 / RPI];
#pragma unroll
        for (int t = 0; t < 32 / RPI; ++t) {
            const int m = m0 + wr * WM + i * 32 + t * RPI + lane / C4;
            rv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (use_r && n_ok && m < p.M) rv[t] = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + n);
        }
#pragma unroll
        for (int t = 0; t < 32 / RPI; ++t) {
            const int row = t * RPI + lane / C4;
            const int m = m0 + wr * WM + i * 32 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * ES + c4);
            v = v * sc + sh + rv[t];
            v[0] = fmaxf(v[0], relu_lo); v[1] = fmaxf(v[1], relu_lo);
            v[2] = fmaxf(v[2], relu_lo); v[3] = fmaxf(v[3], relu_lo);
            if (n_ok && m < p.M) {
                if (p.C) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
                if (p.Cp) {
                    unsigned int a0, a1, a2, b0, b1, b2;
                    split2(v[0], v[1], a0, a1, a2);
                    split2(v[2], v[3], b0, b1, b2);
                    unsigned short* d = p.Cp + (size_t)m * p.ldcp + n;
                    *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
                    *reinterpret_cast<u32x2*>(d + p.c_plane_stride) = u32x2{a1, b1};
                    *reinterpret_cast<u32x2*>(d + 2 * p.c_plane_stride) = u32x2{a2, b2};
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // reads done before the next slab overwrites
    }
}

// fp32 rows [M, ld_in] -> three bf16 planes [3][M][ld_out] (columns K..ld_out zeroed): the stand-alone producer for
// activations whose kernel has no plane epilogue (one pass: 4 B read + 6 B written per element).
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, long ld_in, long M, int K,
                                                         unsigned short* __restrict__ out, int ld_out,
                                                         long plane_stride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // quad index
    const int q_per_row = ld_out / 4;
    if (i >= M * q_per_row) return;
    const long m = i / q_per_row;
    const int k = (int)(i % q_per_row) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < K) v = *reinterpret_cast<const f32x4*>(X + m * ld_in + k);
    unsigned int a0, a1, a2, b0, b1, b2;
    split2(v[0], v[1], a0, a1, a2);
    split2(v[2], v[3], b0, b1, b2);
    unsigned short* d = out + m * ld_out + k;
    *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
    *reinterpret_cast<u32x2*>(d + plane_stride) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(d + 2 * plane_stride) = u32x2{a2, b2};
}

}  // namespace

extern "C" int gom_split_rows_bf16x3(const float* X, long ld_in, long M, int K, void* planes_out, int ld_out,
                                     long plane_stride, void* stream) {
    GOM_CHECK_ARG(X && planes_out && M >= 0 && K > 0 && (K % 4) == 0 && ld_in >= K && (ld_in % 4) == 0);
    GOM_CHECK_ARG(ld_out >= K && (ld_out % 32) == 0 && plane_stride >= M * ld_out && (plane_stride % 8) == 0);
    GOM_CHECK_ARG(((uintptr_t)X % 16) == 0 && ((uintptr_t)planes_out % 16) == 0);
    if (M == 0) return GOM_OK;
    const long quads = M * (ld_out / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, (hipStream_t)stream, X, ld_in,
                       M, K, (unsigned short*)planes_out, ld_out, plane_stride);
    return gom_launch_status();
}

extern "C" int gom_gemm_planes_bf16x6(const void* Aplanes, long a_plane_stride, int lda, const void* Wplanes,
                                      long w_plane_stride, int ldw, const float* scale, const float* shift,
                                      const float* R, int ldr, int r_cols, int relu, float* C, int ldc, void* Cplanes,
                                      long c_plane_stride, int ldcp, int M, int N, int K, void* stream) {
    GOM_CHECK_ARG(Aplanes && Wplanes && (C || Cplanes));
    GOM_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 32) == 0 && (N % 4) == 0);
    GOM_CHECK_ARG((lda % 8) == 0 && lda >= K && (ldw % 32) == 0 && ldw >= K);
    GOM_CHECK_ARG((a_plane_stride % 8) == 0 && (w_plane_stride % 8) == 0);
    GOM_CHECK_ARG(!C || (ldc >= N && (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0));
    GOM_CHECK_ARG(!Cplanes || (ldcp >= N && (ldcp % 4) == 0 && (c_plane_stride % 4) == 0 && ((uintptr_t)Cplanes % 8) == 0));
    GOM_CHECK_ARG(!R || (r_cols > 0 && r_cols <= N && ldr >= r_cols && (ldr % 4) == 0 && (r_cols % 4) == 0));
    GOM_CHECK_ARG(((uintptr_t)Aplanes % 16) == 0 && ((uintptr_t)Wplanes % 16) == 0);
    GOM_CHECK_ARG((long)M * lda * 2 < (1L << 31) && (long)N * ldw * 2 < (1L << 31));
    if (M == 0) return GOM_OK;
    PArgs a{};
    a.Ap = (const unsigned short*)Aplanes; a.Wp = (const unsigned short*)Wplanes;
    a.a_plane_stride = a_plane_stride; a.w_plane_stride = w_plane_stride;
    a.C = C; a.Cp = (unsigned short*)Cplanes; a.c_plane_stride = c_plane_stride;
    a.scale = scale; a.shift = shift; a.R = R;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldcp = ldcp; a.ldr = ldr;
    a.relu = relu; a.r_cols = r_cols;
    // 256x128 / 128x256 tiles with 8 waves (4 waves/SIMD) were measured too: within +-4 % of this one on every shape
    const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    hipLaunchKernelGGL((gemm_planes_kernel<128, 128, 2, 2, 3>), dim3((unsigned)tiles), dim3(256), 3 * (128 + 128) * ROWB,
                       (hipStream_t)stream, a);
    return gom_launch_status();
}
