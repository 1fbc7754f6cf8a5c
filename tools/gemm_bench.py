"""Per-shape throughput of gom_gemm_f32 / gom_conv2d_nhwc_f32 on the shapes of the bench workload
(B = 8 frames of 1000x1778).  Usage (GPU box): python tools/gemm_bench.py [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402

S8 = 37171 * 8
Q8 = 2500 * 8
GEMMS = [
    ("enc raw (src+pos)", S8, 384, 256, True), ("enc value/out", S8, 256, 256, False),
    ("enc ffn1", S8, 1024, 256, False), ("enc ffn2", S8, 256, 1024, False), ("dec values x6", S8, 1536, 256, False),
    ("enc class N=1", S8, 1, 256, False),
    ("dec qk (A2)", Q8, 512, 256, True), ("dec 256x256", Q8, 256, 256, False), ("dec qkv inter", Q8, 768, 256, False),
    ("dec raw", Q8, 384, 256, True), ("dec ffn1", Q8, 1024, 256, False), ("dec ffn2", Q8, 256, 1024, False),
    ("text head", Q8, 38, 256, False), ("coord N=2", Q8, 2, 256, False),
    ("res2 1x1 64->64", 890000, 64, 64, False), ("res2 1x1 64->256", 890000, 256, 64, False),
    ("res2 1x1 256->64", 890000, 64, 256, False), ("res3 1x1 128->512", 222500, 512, 128, False),
    ("res3 1x1 512->128", 222500, 128, 512, False), ("res4 1x1 256->1024", 56448, 1024, 256, False),
    ("res4 1x1 1024->256", 56448, 256, 1024, False), ("res5 1x1 512->2048", 14336, 2048, 512, False),
    ("res5 1x1 2048->512", 14336, 512, 2048, False), ("proj res3", 223000, 256, 512, False),
    ("fc1 6400->1024 (M=200)", 200, 1024, 6400, False),
]
CONVS = [("stem 7x7", 8, 1000, 1778, 4, 64, 7, 2, 3), ("res2 3x3", 8, 250, 445, 64, 64, 3, 1, 1),
         ("res3 3x3 s2", 8, 250, 445, 128, 128, 3, 2, 1), ("res3 3x3", 8, 125, 223, 128, 128, 3, 1, 1),
         ("res4 3x3", 8, 63, 112, 256, 256, 3, 1, 1), ("res5 3x3", 8, 32, 56, 512, 512, 3, 1, 1),
         ("res3 sc 1x1 s2", 8, 250, 445, 256, 512, 1, 2, 0)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--mode", default="f16x3", choices=["f16x3", "bf16x6", "fp32"])
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    a = ap.parse_args()
    ops.GEMM_MODE = a.mode
    dev = "cuda"
    tot_t = tot_f = 0.0
    for name, M, N, K, a2 in GEMMS:
        if a.only and a.only not in name:
            continue
        A = torch.randn(M, K, device=dev)
        A2 = torch.randn(M, K, device=dev) if a2 else None
        W = ops.prep_weight(torch.randn(N, K, device=dev))
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        t = timeit(lambda: ops.gemm(A, W, bias=b, A2=A2, out=out), a.iters)
        fl = 2.0 * M * N * K
        tot_t += t
        tot_f += fl
        print("%-26s M=%7d N=%5d K=%5d  %8.1f us  %6.1f TF" % (name, M, N, K, t * 1e6, fl / t / 1e12))
    for name, B, H, W_, Cin, Cout, k, s, p in CONVS:
        if a.only and a.only not in name:
            continue
        x = torch.randn(B, H, W_, Cin, device=dev)
        w = ops.prep_conv_weight(torch.randn(Cout, k, k, Cin, device=dev))
        sc, sh = torch.rand(Cout, device=dev), torch.randn(Cout, device=dev)
        t = timeit(lambda: ops.conv2d_nhwc(x, w, scale=sc, shift=sh, relu=True, stride=s, pad=p), a.iters)
        OH, OW = (H + 2 * p - k) // s + 1, (W_ + 2 * p - k) // s + 1
        fl = 2.0 * B * OH * OW * Cout * k * k * Cin
        print("%-26s M=%7d N=%5d K=%5d  %8.1f us  %6.1f TF" % (name, B * OH * OW, Cout, k * k * Cin, t * 1e6,
                                                             fl / t / 1e12))
    if tot_t:
        print("GEMM list aggregate: %.1f TF" % (tot_f / tot_t / 1e12))


if __name__ == "__main__":
    main()
