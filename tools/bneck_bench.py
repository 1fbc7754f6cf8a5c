"""Interleaved A/B on one GPU: a bottleneck block's tail fused with the next block's head (csrc/bneck_fused.hip) against the two
launches of the tile kernel, at the ResNet-50 shapes of the bench (8 frames of 1000 x 1778)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(0)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for k1, mp, hw in ((256, 256, (63, 112)), (128, 256, (125, 223)), (128, 128, (125, 223))):
    c4, B = 4 * k1, 8
    a = torch.randn(B, hw[0], hw[1], k1, generator=g).abs().to(DEV)
    R = torch.randn(B, hw[0], hw[1], c4, generator=g).to(DEV)
    w3 = (torch.randn(c4, 1, 1, k1, generator=g) / k1 ** 0.5).to(DEV)
    w1 = (torch.randn(mp, 1, 1, c4, generator=g) / c4 ** 0.5).to(DEV)
    sc3, sh3 = torch.ones(c4, device=DEV), torch.zeros(c4, device=DEV)
    sc1, sh1 = torch.ones(mp, device=DEV), torch.zeros(mp, device=DEV)
    s3 = ops.split_weight(w3.reshape(c4, k1), conv_shape=tuple(w3.shape), kind="f16x3")
    s1 = ops.split_weight(w1.reshape(mp, c4), conv_shape=tuple(w1.shape), kind="f16x3")
    blk = ops.BneckFused(s3, sc3, sh3, s1, sc1, sh1)
    M = B * hw[0] * hw[1]

    def two():
        x = ops.conv2d_nhwc(a, s3, scale=sc3, shift=sh3, R=R, relu=True)
        return ops.conv2d_nhwc(x, s1, scale=sc1, shift=sh1, relu=True)

    def c3_only():
        return ops.conv2d_nhwc(a, s3, scale=sc3, shift=sh3, R=R, relu=True)

    def fused():
        return ops.bneck_fused(a, blk, R)

    t2, t3, tf = timeit(two), timeit(c3_only), timeit(fused)
    t2b, tfb = timeit(two), timeit(fused)
    gb = 4.0 * M * (k1 + 2 * c4 + mp) / 1e9
    print("%3d -> %4d -> %3d, M = %6d: two launches %.0f / %.0f us (conv3 alone %.0f), fused %.0f / %.0f us = %.2f TB/s of %.2f GB, "
          "%.0f TFLOP/s" % (k1, c4, mp, M, t2, t2b, t3, tf, tfb, gb / tfb * 1e-3 * 1e3, gb, 2.0 * M * c4 * (k1 + mp) / tfb / 1e6))
