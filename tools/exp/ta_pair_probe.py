"""would a HEAD-MAJOR value layout make the MSDA gather cheaper for the texture-address path?

The fused MSDA kernel (csrc/msda.hip) reads, per query-wave and (level, point) sample, the 4 bilinear corners of 8 heads: 4
wave-instructions of 16 bytes per lane, each touching EIGHT scattered 128-byte lines (value is [pixel][head][32 floats], every
head samples its own location).  With value stored [head][pixel][32] the two x-adjacent corners of a head are ADJACENT lines,
so an instruction can fetch 4 heads x (2 adjacent lines = one 256-byte run) instead: same bytes, same number of
instructions, half as many distinct runs.  This probe replays the SAME synthetic sampling pattern (one 160 x 232 level,
queries in raster order, 16 samples per query within +-4 px of the query's pixel, per-head random) through both address shapes
with a kernel that does nothing but the loads:
    A  pixel-major, 8 lines per instruction (the shipped kernel's shape)
    B  head-major, 4 x 2 adjacent lines per instruction
    C  head-major, but one line per 8 lanes in the order of A (isolates the layout's cache effect from the run shape)
    python tools/exp/ta_pair_probe.py --build   (here)        python tools/exp/ta_pair_probe.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libta_pair_probe.so")
SRC = r'''
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// idx: [waves][64 instructions][GROUPS] line numbers; lane l reads 16 bytes at line idx[l / LPG] * 128 + (l % LPG) * 16
template <int GROUPS>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ table, const int* __restrict__ idx, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int LPG = 64 / GROUPS;
    const int* my = idx + wave * 64 * GROUPS;
    float acc = 0.f;
#pragma unroll 16
    for (int i = 0; i < 64; ++i) {
        const int line = my[i * GROUPS + lane / LPG];
        const f4 v = *reinterpret_cast<const f4*>(table + (long)line * 32 + (lane % LPG) * 4);
        acc += v[0] + v[1] + v[2] + v[3];
    }
    out[wave * 64 + lane] = acc;
}
extern "C" int run(int groups, const float* table, const int* idx, float* out, int waves) {
    dim3 g(waves / 4), b(256);
    if (groups == 8) hipLaunchKernelGGL(probe<8>, g, b, 0, 0, table, idx, out);
    else hipLaunchKernelGGL(probe<4>, g, b, 0, 0, table, idx, out);
    return (int)hipGetLastError();
}
'''


def build():
    src = os.path.join(HERE, "_ta_pair_probe.hip")
    open(src, "w").write(SRC)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", SO])
    os.remove(src)
    print("built", SO)


def main():
    import torch
    so = ctypes.CDLL(SO)
    so.run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    dev = "cuda"
    H, W = 160, 232
    S = H * W
    frames = 4
    waves = frames * S                                  # one query per pixel and frame, raster order (as the encoder's launches)
    waves -= waves % 4
    g = torch.Generator().manual_seed(0)
    q = torch.arange(waves) % S
    qy, qx = (q // W).view(-1, 1, 1), (q % W).view(-1, 1, 1)
    frame = (torch.arange(waves) // S).view(-1, 1, 1)
    # 16 samples x 8 heads: top-left corner of the bilinear footprint
    x0 = (qx + torch.randint(-4, 4, (waves, 16, 8), generator=g)).clamp(0, W - 2)
    y0 = (qy + torch.randint(-4, 4, (waves, 16, 8), generator=g)).clamp(0, H - 2)
    head = torch.arange(8).view(1, 1, 8)
    table = torch.randn(frames * S * 8 * 32, device=dev)

    def pix(dy, dx):
        return (y0 + dy) * W + x0 + dx

    # A: pixel-major lines (frame, pixel, head); instruction = (sample, corner), 8 groups = heads
    A = torch.stack([((frame * S + pix(dy, dx)) * 8 + head) for dy in (0, 1) for dx in (0, 1)], 2)        # [waves,16,4,8]
    # C: head-major lines (frame, head, pixel), same instruction shape as A
    C = torch.stack([((frame * 8 + head) * S + pix(dy, dx)) for dy in (0, 1) for dx in (0, 1)], 2)
    # B: head-major; instruction = (sample, dy, half of the heads), 4 groups = heads, each a 256-byte run (x0, x0 + 1)
    Bm = torch.stack([((frame * 8 + head[..., 4 * hh:4 * hh + 4]) * S + pix(dy, 0)[..., 4 * hh:4 * hh + 4])
                      for dy in (0, 1) for hh in (0, 1)], 2)                                                  # [waves,16,4,4]
    out = torch.empty(waves * 64, device=dev)
    res = {}
    for name, idx, groups in (("A pixel-major, 8 scattered lines / instruction", A, 8),
                              ("B head-major, 4 x 2 adjacent lines / instruction", Bm, 4),
                              ("C head-major, 8 scattered lines / instruction", C, 8)) * 2:
        d = idx.reshape(waves, 64 * groups).to(torch.int32).contiguous().to(dev)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert so.run(groups, table.data_ptr(), d.data_ptr(), out.data_ptr(), waves) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t = sorted(ts)[2]
        lines = waves * 64 * 8
        res[name] = t
        print("%-52s %8.1f us  %6.2f TB/s of lines  %.2f cycles per 128-B line and CU (2.1 GHz)"
              % (name, t, lines * 128.0 / t / 1e6, 256 * 2.1e9 * t * 1e-6 / lines))
    a = res["A pixel-major, 8 scattered lines / instruction"]
    b = res["B head-major, 4 x 2 adjacent lines / instruction"]
    print("B / A = %.3f  (the layout change pays if <= 0.85)" % (b / a))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
