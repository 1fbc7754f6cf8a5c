"""Diagnostic build of the patch-resident 3x3 convolution (csrc/conv3x3_patch.hip) with s_memtime stamps: per wave, cycles in the
k-loop, in the stage waits + barriers, in the LDS-DMA issue, and to the end of the epilogue.  The product kernel carries no stamps.
    python tools/exp/conv3_clock.py --build   (here: writes tools/exp/libconv3_clock.so = the product library with the stamped
    kernel linked in its place)          python tools/exp/conv3_clock.py   (GPU box)"""
import ctypes
import math
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libconv3_clock.so")


def build():
    csrc = os.path.join(ROOT, "gomatching_amd", "csrc")
    s = open(os.path.join(csrc, "conv3x3_patch.hip")).read()

    def once(s, a, b):
        assert s.count(a) == 1, a
        return s.replace(a, b)
    s = once(s, '#include "common.h"', '#include "common.h"\n__device__ unsigned long long g_dbg[8192 * 4];')
    s = once(s, '    auto stage_sync = [&](int st) {', '    unsigned long long t_sync = 0, t_dma = 0;\n    auto stage_sync = [&](int st) {\n'
             '        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();')
    s = once(s, '        fresh = false;\n        if ((st + 1) * KPS < total) dma_W(st + 1);\n    };',
             '        fresh = false;\n        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();\n'
             '        if ((st + 1) * KPS < total) dma_W(st + 1);\n        t_sync += t1_ - t0_; t_dma += __builtin_amdgcn_s_memtime() - t1_;\n    };\n'
             '    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();')
    s = once(s, '    // ---- epilogue: 32-pixel slabs', '    const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();\n    // ---- epilogue: 32-pixel slabs')
    s = once(s, '    if (bad && p.flag) atomicOr(p.flag, 1);\n}', '    if (bad && p.flag) atomicOr(p.flag, 1);\n'
             '    if (blockIdx.x < 2048 && lane == 0) {\n        unsigned long long* d = g_dbg + (blockIdx.x * 4 + wave) * 4;\n'
             '        d[0] = t_loop_end - t_begin; d[1] = t_sync; d[2] = t_dma; d[3] = __builtin_amdgcn_s_memtime() - t_begin;\n    }\n}')
    s += ('\nextern "C" int gom_conv3_dbg_read(unsigned long long* host) {\n'
          '    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 8192 * 4);\n}\n')
    gen = os.path.join(csrc, "_conv3_clock_gen.hip")
    open(gen, "w").write(s)
    obj = os.path.join(HERE, "_conv3_clock.o")
    try:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-x", "hip", "-c", gen, "-o", obj])
    finally:
        os.remove(gen)
    od = os.path.join(csrc, "_obj")
    objs = [os.path.join(od, f) for f in sorted(os.listdir(od)) if f.endswith(".o") and f != "conv3x3_patch.hip.o"]
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", SO] + objs + [obj])
    os.remove(obj)
    print("built", SO)


def main():
    os.environ["GOM_LIB_PATH"] = SO
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from gomatching_amd import ops, lib
    dev = "cuda"
    ops.GEMM_MODE = "f16x3"
    g = torch.Generator().manual_seed(0)
    L = lib.load()
    L.gom_conv3_dbg_read.argtypes = [ctypes.c_void_p]
    for (H, W, C) in [(250, 445, 64), (125, 223, 128), (63, 112, 256), (32, 56, 512)]:
        x = torch.randn((8, H, W, C), generator=g).to(dev)
        w = (torch.randn((C, 3, 3, C), generator=g) / math.sqrt(9 * C)).to(dev)
        sw = ops.split_weight(w.reshape(C, -1), conv_shape=tuple(w.shape), kind="f16x3")
        for _ in range(20):
            ops.conv2d_nhwc(x, sw, relu=True, stride=1, pad=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d_nhwc(x, sw, relu=True, stride=1, pad=1)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        buf = np.zeros((2048, 4, 4), dtype=np.uint64)
        assert L.gom_conv3_dbg_read(buf.ctypes.data) == 0
        wgs = 8 * ((H + 7) // 8) * ((W + 15) // 16) * (C // (128 if C % 128 == 0 else 64))
        b = buf[:min(wgs, 2048)].astype(np.int64)
        kts = (C // 64) * 18
        floor = (48 if C % 128 == 0 else 24) * 16
        med = lambda a: float(np.median(a))
        rounds = wgs / 512.0
        print("C %3d: %6.1f us per launch; per wave, median cycles: k-loop %7.0f = %5.0f per k-tile (own MFMAs %d, two waves per SIMD: "
              "%.0f %% pipe busy) | waits + barriers %4.0f | DMA issue %4.0f per k-tile | whole kernel %7.0f; %.2f rounds of 512 "
              "workgroups -> %.2f GHz" % (C, us, med(b[:, :, 0]), med(b[:, :, 0]) / kts, floor, 200.0 * floor * kts / med(b[:, :, 0]),
                                          med(b[:, :, 1]) / kts, med(b[:, :, 2]) / kts, med(b[:, :, 3]), rounds,
                                          math.ceil(rounds) * med(b[:, :, 3]) / us / 1e3))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
