"""A/B in ONE process: bf16x6 vs f16x3 GEMM (fp32 A, pre-split W) on the bench shapes: time and error vs fp64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402

S8 = 37171 * 8
SHAPES = [("enc fused N=640", S8, 640, 256), ("enc out_proj", S8, 256, 256), ("enc ffn1", S8, 1024, 256),
          ("enc ffn2", S8, 256, 1024), ("dec values N=1536", S8, 1536, 256), ("res2 1x1 64->256", 890000, 256, 64),
          ("dec M=20000 N=256", 20000, 256, 256), ("dec M=20000 N=1024", 20000, 1024, 256)]


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


def main(rounds=3, iters=5):
    dev = "cuda"
    for name, M, N, K in SHAPES:
        A = torch.randn(M, K, device=dev)
        Wf = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        W6, W3 = ops.split_weight(Wf, kind="bf16x6"), ops.split_weight(Wf, kind="f16x3")
        out = torch.empty(M, N, device=dev)
        rows = slice(0, 4096)
        ref = (A[rows].double() @ Wf.double().t() + b.double())
        e6 = (ops.gemm(A[rows], W6, bias=b).double() - ref).abs().max().item()
        e3 = (ops.gemm(A[rows], W3, bias=b).double() - ref).abs().max().item()
        e32 = ((A[rows] @ Wf.t() + b).double() - ref).abs().max().item()
        t6, t3 = [], []
        for _ in range(rounds):
            t6.append(timeit(lambda: ops.gemm(A, W6, bias=b, out=out), iters))
            t3.append(timeit(lambda: ops.gemm(A, W3, bias=b, out=out), iters))
        fl = 2.0 * M * N * K
        a, c = min(t6), min(t3)
        print("%-20s M=%7d N=%5d K=%5d  bf16x6 %8.1f us %6.1f TF | f16x3 %8.1f us %6.1f TF  x%.2f | max err: bf16x6 %.1e "
              "f16x3 %.1e torch-fp32 %.1e" % (name, M, N, K, a * 1e6, fl / a / 1e12, c * 1e6, fl / c / 1e12, a / c, e6, e3, e32))
    ops.check_range_flag(torch.device(dev))
    big = torch.full((256, 256), 70000.0, device=dev)
    ops.gemm(big, ops.split_weight(torch.ones(64, 256, device=dev), kind="f16x3"))
    try:
        ops.check_range_flag(torch.device("cuda:0") if False else big.device)
        print("range flag NOT raised (unexpected)")
    except Exception as e:
        print("range flag raised as expected:", str(e)[:60])


if __name__ == "__main__":
    main()
