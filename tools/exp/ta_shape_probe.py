"""How many bytes per clock does the vector-memory path of a CU deliver for a GATHER of 128-byte lines, as a function of the
load width -- 16 bytes per lane (8 lanes per line, 8 lines per wave-instruction: the fused MSDA kernel's corner loads), 8 bytes
(16 lanes per line, 4 lines) or 4 bytes (32 lanes per line, 2 lines)?  Lines are drawn from a 38 MB table (an encoder value map)
with the locality of neighbouring queries (consecutive waves read lines near each other).
    python tools/exp/ta_shape_probe.py --build   (here)        python tools/exp/ta_shape_probe.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libta_shape_probe.so")
SRC = r'''
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
// every wave: ITER rounds; in a round it gathers LINES_PER_ROUND lines of 128 B whose indices come from idx (precomputed, local)
template <int W>   // bytes per lane: 16, 8, 4
__global__ __launch_bounds__(256) void probe(const float* __restrict__ table, const int* __restrict__ idx, float* __restrict__ out,
                                             int rounds) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int LPL = 128 / W;                 // lanes per line
    constexpr int LPI = 64 / LPL;                // lines per instruction
    const int* my = idx + wave * rounds * 64;    // 64 line indices per round
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const int* ix = my + r * 64;
#pragma unroll
        for (int i = 0; i < 64 / LPI; ++i) {     // 64 lines per round -> 64 / LPI instructions
            const int line = ix[i * LPI + lane / LPL];
            const float* p = table + (long)line * 32 + (lane % LPL) * (W / 4);
            if constexpr (W == 16) { const f4 v = *reinterpret_cast<const f4*>(p); acc += v[0] + v[1] + v[2] + v[3]; }
            else if constexpr (W == 8) { const f2 v = *reinterpret_cast<const f2*>(p); acc += v[0] + v[1]; }
            else acc += *p;
        }
    }
    out[wave * 64 + lane] = acc;
}
extern "C" int run(int w, const float* table, const int* idx, float* out, int waves, int rounds) {
    dim3 g(waves / 4), b(256);
    if (w == 16) hipLaunchKernelGGL(probe<16>, g, b, 0, 0, table, idx, out, rounds);
    else if (w == 8) hipLaunchKernelGGL(probe<8>, g, b, 0, 0, table, idx, out, rounds);
    else hipLaunchKernelGGL(probe<4>, g, b, 0, 0, table, idx, out, rounds);
    return (int)hipGetLastError();
}
'''


def build():
    src = os.path.join(HERE, "_ta_shape_probe.hip")
    open(src, "w").write(SRC)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", SO])
    os.remove(src)
    print("built", SO)


def main():
    import torch
    so = ctypes.CDLL(SO)
    so.run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    dev = "cuda"
    lines = 37171 * 8                                  # one frame's value map: 37 171 pixels x 8 head-lines of 128 B
    table = torch.randn(lines * 32, device=dev)
    waves, rounds = 256 * 32 * 8, 8                    # 8 waves per SIMD-slot set, 8 rounds of 64 lines = 512 lines per wave
    g = torch.Generator().manual_seed(0)
    # locality of neighbouring queries: wave w reads around pixel (w * 37171 / waves) within +-400 pixels, any head-line
    centre = (torch.arange(waves).double() * (37171.0 / waves)).long().view(-1, 1, 1)
    pix = (centre + torch.randint(-400, 400, (waves, rounds, 64), generator=g)).clamp(0, 37170)
    idx = (pix * 8 + torch.randint(0, 8, (waves, rounds, 64), generator=g)).to(torch.int32).to(dev)
    out = torch.empty(waves * 64, device=dev)
    for w in (16, 8, 4, 16, 8, 4):
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert so.run(w, table.data_ptr(), idx.data_ptr(), out.data_ptr(), waves, rounds) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts)[2]
        nbytes = waves * rounds * 64 * 128.0
        print("%2d bytes per lane: %7.1f us for %.1f GB of lines = %5.2f TB/s = %.1f B per clock and CU (2.1 GHz), %.2f cycles per line"
              % (w, t, nbytes / 1e9, nbytes / t / 1e6, nbytes / t / 1e6 * 1e12 / 256 / 2.1e9, 256 * 2.1e9 * t * 1e-6 / (waves * rounds * 64)))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
