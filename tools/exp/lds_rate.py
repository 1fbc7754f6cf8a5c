"""What does the LDS deliver to conflict-free `ds_read_b128` reads of a fragment-linear image (lane l reads bytes 16 l .. 16 l + 15
of a 1 KB fragment: the layout every row-resident kernel here uses for its weights)?  MI355X_MICROARCH.md (LDS) says 256 B/clk/CU;
rounds 3-5 of this repo sized their decoder arithmetic against 128.  One workgroup per CU (96 KB of LDS requested), 4 or 8 waves:

  pure   : 16 reads in flight, `s_waitcnt lgkmcnt(0)`, repeat               -> B/clk/CU of the LDS alone
  mix R:G: per group R reads (the NEXT group's fragments) + G `v_mfma_f32_16x16x32_f16` on the current ones, random operands;
           2:6 = the fused FFN / decoder tail today (32 rows per wave), 2:3 = a 16-row wave, 2:15 = an 80-row split-N wave
           -> cycles per group against G x 16 (8 passes x 4 cycles x ... = 16 issue cycles per MFMA at full rate)

Cycles: s_memtime (shader clock) per wave, median over the launch; wall time beside it gives the clock the chip held.
    python tools/exp/lds_rate.py --build   (here)        python tools/exp/lds_rate.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liblds_rate.so")
SRC = r'''
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int FRAGS = 64;

__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void pure_reads(const u32x4* __restrict__ src, unsigned* __restrict__ out,
                                                          unsigned long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* w = reinterpret_cast<u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < FRAGS * 64; i += WAVES * 64) w[i] = src[i];
    __syncthreads();
    const unsigned addr = lane * 16;
    u32x4 v[16];
    const unsigned long long t0 = memtime();
    for (int it = 0; it < iters; ++it) {
#define RD(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(((i) * 4 + 1) * 1024));
        RD(0) RD(1) RD(2) RD(3) RD(4) RD(5) RD(6) RD(7) RD(8) RD(9) RD(10) RD(11) RD(12) RD(13) RD(14) RD(15)
#undef RD
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = memtime();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + (tid >> 6)] = t1 - t0;
}

// R reads + G MFMAs per group, software-pipelined one group deep; 8 groups per loop trip
template <int WAVES, int R, int G>
__global__ __launch_bounds__(WAVES * 64, 1) void mix(const half8* __restrict__ src, float* __restrict__ out,
                                                    unsigned long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8* w = reinterpret_cast<half8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < FRAGS * 64; i += WAVES * 64) w[i] = src[i];
    half8 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = src[(FRAGS + (tid >> 6) * 4 + i) * 64 + lane];
    __syncthreads();
    f32x4 acc[8] = {};
    half8 cur[R], nxt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = w[r * 64 + lane];
    const unsigned long long t0 = memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int r = 0; r < R; ++r) nxt[r] = w[(((g + 1) * R + r) & (FRAGS - 1)) * 64 + lane];
#pragma unroll
            for (int j = 0; j < G; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[j % R], x[j & 3], acc[j & 7], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
#pragma unroll
            for (int r = 0; r < R; ++r) cur[r] = nxt[r];
        }
    }
    const unsigned long long t1 = memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + (tid >> 6)] = t1 - t0;
}

template <typename K>
static int launch(K k, int waves, const void* src, void* out, unsigned long long* cyc, int blocks, int iters) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), 96 * 1024, 0, (decltype(src))src, (decltype(out))out, cyc, iters);
    return (int)hipGetLastError();
}

extern "C" int run_pure(int waves, const void* src, void* out, unsigned long long* cyc, int blocks, int iters) {
    if (waves == 4) { hipFuncSetAttribute((const void*)pure_reads<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL((pure_reads<4>), dim3(blocks), dim3(256), 96 * 1024, 0, (const u32x4*)src, (unsigned*)out, cyc, iters); }
    else if (waves == 8) { hipFuncSetAttribute((const void*)pure_reads<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL((pure_reads<8>), dim3(blocks), dim3(512), 96 * 1024, 0, (const u32x4*)src, (unsigned*)out, cyc, iters); }
    else { hipFuncSetAttribute((const void*)pure_reads<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL((pure_reads<16>), dim3(blocks), dim3(1024), 96 * 1024, 0, (const u32x4*)src, (unsigned*)out, cyc, iters); }
    return (int)hipGetLastError();
}
#define MIX(W, R, G)                                                                                                         \
    if (waves == W && r == R && g == G) {                                                                                    \
        hipFuncSetAttribute((const void*)mix<W, R, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);               \
        hipLaunchKernelGGL((mix<W, R, G>), dim3(blocks), dim3(W * 64), 96 * 1024, 0, (const half8*)src, (float*)out, cyc, iters); \
        return (int)hipGetLastError();                                                                                       \
    }
extern "C" int run_mix(int waves, int r, int g, const void* src, void* out, unsigned long long* cyc, int blocks, int iters) {
    MIX(4, 2, 6) MIX(8, 2, 6) MIX(4, 2, 3) MIX(8, 2, 3) MIX(4, 2, 15) MIX(8, 2, 15) MIX(4, 2, 2) MIX(8, 2, 2) MIX(4, 2, 1) MIX(8, 2, 1)
    MIX(4, 2, 4) MIX(8, 2, 4) MIX(4, 2, 10) MIX(4, 0 + 1, 0 + 1) MIX(8, 1, 1)
    return -1;
}
'''


def build():
    src = os.path.join(HERE, "_lds_rate.hip")
    open(src, "w").write(SRC)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", SO])
    os.remove(src)
    print("built", SO)


def main():
    import torch
    so = ctypes.CDLL(SO)
    dev = "cuda"
    blocks = 256
    src = (torch.randn((64 + 64) * 64 * 8, device=dev) * 0.5).half()
    out = torch.empty(blocks * 1024, device=dev)
    cyc = torch.zeros(blocks * 16, dtype=torch.int64, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n          # us

    print("== pure ds_read_b128, fragment-linear (conflict-free), one workgroup per CU, 256 workgroups ==")
    for waves in (4, 8, 16):
        iters = 20000
        us = timed(lambda: so.run_pure(waves, p(src), p(out), p(cyc), blocks, iters))
        # (the waves of a SIMD do not take equal turns -- the older one runs ahead -- so a workgroup's time is its SLOWEST wave's)
        c = cyc[:blocks * waves].view(blocks, waves).float().max(dim=1).values.median().item()
        byt = waves * iters * 16 * 1024
        print("  %2d waves / CU: %8.0f cycles (median workgroup) -> %6.1f B/clk/CU; wall %8.1f us -> %5.2f GHz, %6.1f TB/s chip"
              % (waves, c, byt / c, us, c / us / 1e3, byt * blocks / us / 1e6))
    print("== R reads + G v_mfma_f32_16x16x32_f16 per group (next group's fragments under this group's MFMAs), random operands ==")
    for r, g in ((2, 6), (2, 3), (2, 15), (2, 10), (2, 4), (2, 2), (2, 1), (1, 1)):
        for waves in (4, 8):
            iters = 4000
            rc = so.run_mix(waves, r, g, p(src), p(out), p(cyc), blocks, iters)
            if rc != 0:
                continue
            us = timed(lambda: so.run_mix(waves, r, g, p(src), p(out), p(cyc), blocks, iters))
            c = cyc[:blocks * waves].view(blocks, waves).float().max(dim=1).values.median().item()
            groups = iters * 8
            per_simd_mfma = (waves // 4) * groups * g * 16.0          # issue cycles the SIMD's matrix pipe needs
            print("  %d:%-2d  %d waves / CU: %7.1f cycles per group and wave (MFMA alone %4d) -> pipe busy %5.1f %%, LDS %6.1f B/clk/CU; "
                  "clock %5.2f GHz" % (r, g, waves, c / groups, g * 16, 100.0 * per_simd_mfma / c, waves * groups * r * 1024 / c,
                                       c / us / 1e3))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
