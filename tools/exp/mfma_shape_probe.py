"""Does the fp16 MFMA SHAPE matter at the operating point of this repo's row-resident f16x3 kernels?  MI355X_MICROARCH.md (DVFS
give-back, item 7) reports bare `16x16x32` loops at 1.12-1.15x the FLOP/s of `32x32x16` loops on random data at equal cycles
(the chip holds a higher clock).  This probe repeats that at the fused FFN kernel's mix: one wave per SIMD, the wave's 32 rows
as register-resident operand fragments (128 VGPRs), every weight fragment re-read from LDS by ds_read_b128, three plane
products per k-step (lo x hi, hi x lo, hi x hi), and FILL independent v_fma per k-step standing in for the kernel's conversion /
epilogue VALU work.  Random data everywhere.
    python tools/exp/mfma_shape_probe.py --build   (here)        python tools/exp/mfma_shape_probe.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmfma_shape_probe.so")
SRC = r'''
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDS_FRAGS = 64;                   // 64 KB of weight fragments, walked round and round

// SHAPE 0: v_mfma_f32_32x32x16_f16, one 32x32 accumulator tile per 32 output columns, 16 k-steps of 16
// SHAPE 1: v_mfma_f32_16x16x32_f16, 2 column groups x 2 row groups of 16x16 per 32 output columns, 8 k-steps of 32
template <int SHAPE, int FILL>
__global__ __launch_bounds__(256, 1) void probe(const half8* __restrict__ wsrc, const half8* __restrict__ xsrc,
                                                float* __restrict__ out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8* w = reinterpret_cast<half8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LDS_FRAGS * 64; i += 256) w[i] = wsrc[i];
    half8 x[2][16];                              // [plane][k-step of 16]  or  [plane][row group * 8 + k-step of 32]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int s = 0; s < 16; ++s) x[p][s] = xsrc[((blockIdx.x * 4 + (tid >> 6)) * 32 + p * 16 + s) * 64 + lane];
    __syncthreads();
    float fill[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    float total = 0.f;
    int f = 0;                                   // fragment cursor in the LDS ring
    for (int t = 0; t < tiles; ++t) {
        if constexpr (SHAPE == 0) {
            f32x16 acc = {0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const half8 whi = w[(f & (LDS_FRAGS - 1)) * 64 + lane], wlo = w[((f + 1) & (LDS_FRAGS - 1)) * 64 + lane];
                f += 2;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, x[1][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, x[0][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, x[0][s], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < FILL; ++i) fill[i & 7] = __builtin_fmaf(fill[i & 7], 1.0001f, 0.5f);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) total += acc[i];
        } else {
            f32x4 acc[2][2] = {};
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const half8 whi = w[(f & (LDS_FRAGS - 1)) * 64 + lane], wlo = w[((f + 1) & (LDS_FRAGS - 1)) * 64 + lane];
                    f += 2;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        acc[c][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, x[1][r * 8 + s], acc[c][r], 0, 0, 0);
                        acc[c][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, x[0][r * 8 + s], acc[c][r], 0, 0, 0);
                        acc[c][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, x[0][r * 8 + s], acc[c][r], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < FILL / 2; ++i) fill[i & 7] = __builtin_fmaf(fill[i & 7], 1.0001f, 0.5f);
                }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r) total += acc[c][r][0] + acc[c][r][1] + acc[c][r][2] + acc[c][r][3];
        }
    }
    float fs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) fs += fill[i];
    out[(size_t)blockIdx.x * 256 + tid] = total + fs;
}

template <int SHAPE, int FILL>
static int go(const void* w, const void* x, float* out, int blocks, int tiles) {
    hipFuncSetAttribute((const void*)probe<SHAPE, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipLaunchKernelGGL((probe<SHAPE, FILL>), dim3(blocks), dim3(256), 96 * 1024, 0, (const half8*)w, (const half8*)x, out, tiles);
    return (int)hipGetLastError();
}
extern "C" int run(int shape, int fill, const void* w, const void* x, float* out, int blocks, int tiles) {
    if (shape == 0) return fill == 0 ? go<0, 0>(w, x, out, blocks, tiles) : fill == 8 ? go<0, 8>(w, x, out, blocks, tiles)
                                                                                     : go<0, 24>(w, x, out, blocks, tiles);
    return fill == 0 ? go<1, 0>(w, x, out, blocks, tiles) : fill == 8 ? go<1, 8>(w, x, out, blocks, tiles)
                                                                      : go<1, 24>(w, x, out, blocks, tiles);
}
'''


def build():
    src = os.path.join(HERE, "_mfma_shape_probe.hip")
    open(src, "w").write(SRC)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", SO])
    os.remove(src)
    print("built", SO)


def main():
    import torch
    so = ctypes.CDLL(SO)
    dev = "cuda"
    blocks, tiles = 256 * 4, 2048                      # 4 rounds of one workgroup per CU; 2048 column tiles of 32 per wave
    w = (torch.randn(64 * 64 * 8, device=dev) * 0.5).half()
    x = (torch.randn(blocks * 4 * 32 * 64 * 8, device=dev) * 0.5).half()
    out = torch.empty(blocks * 256, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    flop = 2.0 * blocks * 4 * tiles * 32 * 32 * 256 * 3          # per wave and tile: 32 rows x 32 cols x K 256 x 3 plane products
    for fill in (0, 8, 24):
        res = {}
        for rep in range(2):
            for shape in (0, 1):
                for _ in range(3):
                    assert so.run(shape, fill, p(w), p(x), p(out), blocks, tiles) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    so.run(shape, fill, p(w), p(x), p(out), blocks, tiles)
                e1.record()
                torch.cuda.synchronize()
                res[shape] = e0.elapsed_time(e1) / 10
            print("fill %2d VALU per k-step: 32x32x16 %7.3f ms (%6.1f TFLOP/s raw)   16x16x32 %7.3f ms (%6.1f)   ratio %.3f"
                  % (fill, res[0], flop / res[0] / 1e9, res[1], flop / res[1] / 1e9, res[0] / res[1]))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
