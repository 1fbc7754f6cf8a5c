"""The bottlenecks' 3x3 / stride 1 convolutions at the bench's sizes: patch-resident kernel (csrc/conv3x3_patch.hip) against the
implicit-GEMM kernel, interleaved in one process.    python tools/conv3x3_bench.py"""
import math
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
ops.GEMM_MODE = "f16x3"
g = torch.Generator().manual_seed(0)
for (H, W, C) in [(250, 445, 64), (125, 223, 128), (63, 112, 256), (32, 56, 512)]:
    B = 8
    x = torch.randn((B, H, W, C), generator=g).to(dev)
    w = (torch.randn((C, 3, 3, C), generator=g) / math.sqrt(9 * C)).to(dev)
    sw = ops.split_weight(w.reshape(C, -1), conv_shape=tuple(w.shape), kind="f16x3")
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = {}
    for rep in range(2):
        for name, flag in (("patch", True), ("implicit", False)):
            ops.CONV3_PATCH = flag
            for _ in range(3):
                ops.conv2d_nhwc(x, sw, scale=sc, shift=sh, relu=True, stride=1, pad=1)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.conv2d_nhwc(x, sw, scale=sc, shift=sh, relu=True, stride=1, pad=1); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res[name] = ts[len(ts) // 2]
    fl = 2.0 * B * H * W * C * 9 * C
    print("%4dx%4d C %3d: patch %7.1f us (%5.1f TFLOP/s, %.3f of 833)   implicit GEMM %7.1f us (%.3f)   x%.2f" % (
        H, W, C, res["patch"], fl / res["patch"] / 1e6, fl / res["patch"] / 1e6 / 833.3, res["implicit"],
        fl / res["implicit"] / 1e6 / 833.3, res["implicit"] / res["patch"]))
