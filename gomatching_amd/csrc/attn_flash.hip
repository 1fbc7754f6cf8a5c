// Full self-attention of the ViTAEv2 stages 3-4 (NormalCell.py:46-58, token_transformer.py:27-44) with the score matrix
// kept on the CU: softmax(scale * Q K^T) V for N up to thousands of tokens and 64- / 128-wide heads, fp32 in / fp32 out,
// both products on the fp16 matrix cores through the same two-plane split as gemm_f16x3.hip (a.b ~ a0b0 + a0b1 + a1b0,
// fp32 accumulation), online softmax in fp32.
//
// One workgroup = 128 queries of one (image, head); wave w owns 32 of them.  Everything is computed TRANSPOSED so that a
// query is a LANE (column of the 32x32 MFMA result) in both products and the softmax bookkeeping is per-lane scalars:
//   S^T[key, query] = scale * K . Q^T      A operand = K tile from LDS, B operand = Q fragments (registers, loaded once)
//   O^T[d,   query] += V^T . P^T           A operand = V^T tile from LDS, B operand = P straight from the S^T accumulators
// The C layout of v_mfma_f32_32x32x16_f16 puts result row (r&3) + 8(r>>2) + 4(lane>>5) in register r; used as the B operand
// of the second product the 8 registers of a k16 step are the keys base + 4(lane>>5) + (j&3) + 8(j>>2), so the V^T
// fragments are read from LDS in that same key order (two 8-byte pieces per lane) -- no shuffle, no LDS round trip for P.
// Row max / sum need one exchange between the two half-waves (lane ^ 32) per key tile.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int QT = 128, KT = 64;                  // queries per workgroup, keys per staged tile

// x = h0 + h1 with h0 = fp16(x), h1 = fp16(x - h0), two elements at a time on the packed converts -- the form
// gemm_f16x3.hip uses.  (A scalar `(_Float16)x` version of this lost the low plane of an element whose fp32 value sat
// exactly on an fp16 rounding tie: one query row off by 4e-5; tools/flash_diag3.py.)
struct Split2 {
    half2_t hi, lo;
};
__device__ __forceinline__ Split2 split2(float x, float y) {
    unsigned int hi, lo;
    gom_split2_f16(x, y, hi, lo);
    return Split2{__builtin_bit_cast(half2_t, hi), __builtin_bit_cast(half2_t, lo)};
}
#define SPLIT2_INTO(x, y, P0, P1, i)        \
    do {                                    \
        const Split2 s2_ = split2((x), (y)); \
        P0[(i)] = s2_.hi[0]; P0[(i) + 1] = s2_.hi[1]; \
        P1[(i)] = s2_.lo[0]; P1[(i) + 1] = s2_.lo[1]; \
    } while (0)

template <int HD>
__global__ __launch_bounds__(256) void flash_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, float* __restrict__ out, int N,
                                                              int ld, int ldo, float scale, int* __restrict__ flag) {
    constexpr int KS = HD + 8;                    // K tile row (halfs): [key][d], 16-byte padded
    constexpr int VS = KT + 8;                    // V^T tile row (halfs): [d][key]
    constexpr int NKS = HD / 16;                  // k16 steps of the first product
    constexpr int MO = HD / 32;                   // 32-row blocks of O^T
    extern __shared__ __attribute__((aligned(16))) unsigned char flash_smem[];     // 36 KB (HD 64) / 71.7 KB (HD 128)
    _Float16(*Ks)[KT * KS] = reinterpret_cast<_Float16(*)[KT * KS]>(flash_smem);
    _Float16(*Vt)[HD * VS] = reinterpret_cast<_Float16(*)[HD * VS]>(flash_smem + 2 * KT * KS * sizeof(_Float16));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
    const int h = blockIdx.y;
    const long b = blockIdx.z;
    const int q0 = blockIdx.x * QT + wave * 32;
    const float* qb = q + (b * N) * (long)ld + h * HD;
    const float* kb = k + (b * N) * (long)ld + h * HD;
    const float* vb = v + (b * N) * (long)ld + h * HD;

    // ---- Q fragments (B operand of S^T = K Q^T): lane = query fr, k-slots = 8 consecutive d per k16 step and half-wave
    half8 qf[2][NKS];
    {
        const int qi = min(q0 + fr, N - 1);       // tail queries recompute the last row, never stored
        const float* qr = qb + (long)qi * ld;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qr + ks * 16 + fh * 8);
            const f32x4 c = *reinterpret_cast<const f32x4*>(qr + ks * 16 + fh * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                SPLIT2_INTO(a[e], a[e + 1], qf[0][ks], qf[1][ks], e);
                SPLIT2_INTO(c[e], c[e + 1], qf[0][ks], qf[1][ks], 4 + e);
            }
        }
    }

    f32x16 oacc[MO];
#pragma unroll
    for (int i = 0; i < MO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int kt = 0; kt < N; kt += KT) {
        __syncthreads();                          // the previous tile's fragments have been read
        // ---- stage K [key][d] and V^T [d][key] as two fp16 planes: each thread owns 4 keys x 4 d blocks
        for (int u = tid; u < (KT / 4) * (HD / 4); u += 256) {
            const int kg = u / (HD / 4), dg = u % (HD / 4);
            f32x4 kr[4], vr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kt + kg * 4 + j;
                kr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                vr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (key < N) {
                    kr[j] = *reinterpret_cast<const f32x4*>(kb + (long)key * ld + dg * 4);
                    vr[j] = *reinterpret_cast<const f32x4*>(vb + (long)key * ld + dg * 4);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {         // K: row = key, 4 consecutive d
                half4 p0, p1;
                SPLIT2_INTO(kr[j][0], kr[j][1], p0, p1, 0);
                SPLIT2_INTO(kr[j][2], kr[j][3], p0, p1, 2);
                *reinterpret_cast<half4*>(&Ks[0][(kg * 4 + j) * KS + dg * 4]) = p0;
                *reinterpret_cast<half4*>(&Ks[1][(kg * 4 + j) * KS + dg * 4]) = p1;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {         // V^T: row = d, 4 consecutive keys
                half4 p0, p1;
                SPLIT2_INTO(vr[0][e], vr[1][e], p0, p1, 0);
                SPLIT2_INTO(vr[2][e], vr[3][e], p0, p1, 2);
                *reinterpret_cast<half4*>(&Vt[0][(dg * 4 + e) * VS + kg * 4]) = p0;
                *reinterpret_cast<half4*>(&Vt[1][(dg * 4 + e) * VS + kg * 4]) = p1;
            }
        }
        __syncthreads();

        // ---- S^T tile: 64 keys x 32 queries per wave
        f32x16 sacc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[mt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const half8 k0 = *reinterpret_cast<const half8*>(&Ks[0][(mt * 32 + fr) * KS + ks * 16 + fh * 8]);
                const half8 k1 = *reinterpret_cast<const half8*>(&Ks[1][(mt * 32 + fr) * KS + ks * 16 + fh * 8]);
                f32x16 c = sacc[mt];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[0][ks], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[1][ks], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[0][ks], c, 0, 0, 0);
                sacc[mt] = c;
            }
        }
        // ---- online softmax over this tile's keys (register r of block mt <-> key kt + 32 mt + (r&3) + 8(r>>2) + 4 fh)
        float mx = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                sacc[mt][r] = key < N ? sacc[mt][r] * scale : -INFINITY;     // (q k^T) * scale, as the reference orders it
                mx = fmaxf(mx, sacc[mt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);     // finite: every tile holds at least one valid key
        const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
        float ps = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[mt][r] = expf(sacc[mt][r] - m_new);
                ps += sacc[mt][r];
            }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < MO; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

        // ---- O^T += V^T P^T: k16 step s uses registers 8(s&1)..+7 of block s>>1 = keys 16 s + 4 fh + (j&3) + 8(j>>2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            half8 p0, p1;
#pragma unroll
            for (int j = 0; j < 8; j += 2)
                SPLIT2_INTO(sacc[s >> 1][8 * (s & 1) + j], sacc[s >> 1][8 * (s & 1) + j + 1], p0, p1, j);
#pragma unroll
            for (int i = 0; i < MO; ++i) {
                const _Float16* r0 = &Vt[0][(i * 32 + fr) * VS + 16 * s + 4 * fh];
                const _Float16* r1 = &Vt[1][(i * 32 + fr) * VS + 16 * s + 4 * fh];
                half8 v0, v1;
                const half4 a0 = *reinterpret_cast<const half4*>(r0), a1 = *reinterpret_cast<const half4*>(r0 + 8);
                const half4 c0 = *reinterpret_cast<const half4*>(r1), c1 = *reinterpret_cast<const half4*>(r1 + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = a0[e]; v0[4 + e] = a1[e];
                    v1[e] = c0[e]; v1[4 + e] = c1[e];
                }
                f32x16 c = oacc[i];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, p0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, p1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, p0, c, 0, 0, 0);
                oacc[i] = c;
            }
        }
    }

    // ---- out[query, d] = O^T / l: register r of block i <-> d = 32 i + (r&3) + 8(r>>2) + 4 fh: float4 pieces per lane
    const int qi = q0 + fr;
    const float inv = 1.f / l_run;
    bool bad = false;
    if (qi < N) {
        float* orow = out + (b * N + qi) * (long)ldo + h * HD;
#pragma unroll
        for (int i = 0; i < MO; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {oacc[i][4 * g] * inv, oacc[i][4 * g + 1] * inv, oacc[i][4 * g + 2] * inv,
                                 oacc[i][4 * g + 3] * inv};
                bad |= !(fabsf(o[0]) <= 3.4e38f && fabsf(o[1]) <= 3.4e38f && fabsf(o[2]) <= 3.4e38f && fabsf(o[3]) <= 3.4e38f);
                *reinterpret_cast<f32x4*>(orow + 32 * i + 8 * g + 4 * fh) = o;
            }
    }
    if (bad && flag) atomicOr(flag, 1);           // an operand left fp16's range (gemm_f16x3.hip's contract)
}

}  // namespace

extern "C" int gom_flash_attention_f32(const float* q, const float* k, const float* v, float* out, int batch, int N, int heads,
                                       int head_dim, int ld, int ldo, int* flag, void* stream) {
    GOM_CHECK_ARG(q && k && v && out && batch > 0 && N > 0 && heads > 0 && (head_dim == 64 || head_dim == 128));
    GOM_CHECK_ARG(ld >= heads * head_dim && ldo >= heads * head_dim && (ld % 4) == 0 && (ldo % 4) == 0);
    GOM_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0);
    GOM_CHECK_ARG((long)batch * N * ld < (1L << 40) && heads < 65536 && batch < 65536);
    const dim3 grid((unsigned)cdiv(N, QT), (unsigned)heads, (unsigned)batch);
    const float scale = 1.0f / sqrtf((float)head_dim);
    const size_t lds = 2 * sizeof(_Float16) * ((size_t)KT * (head_dim + 8) + (size_t)head_dim * (KT + 8));
    if (head_dim == 64) {
        hipLaunchKernelGGL(flash_attention_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, q, k, v, out, N, ld, ldo,
                           scale, flag);
    } else {
        auto kern = flash_attention_kernel<128>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return GOM_ERR_HIP_BASE + (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, q, k, v, out, N, ld, ldo, scale, flag);
    }
    return gom_launch_status();
}
