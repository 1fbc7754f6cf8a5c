"""Which kernel for the matcher's linear layers (K = 1024, N = 1024 / 3072) as the row count grows: one-wave-per-column
VALU (small), exact-fp32 MFMA with deterministic split-K, plain exact-fp32 MFMA, bf16x6 MFMA with pre-split weights."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    for N, K in ((1024, 1024), (3072, 1024), (256, 1024)):
        W = torch.randn(N, K, device=dev)
        Ws = ops.split_weight(W)
        b = torch.randn(N, device=dev)
        for M in (1, 4, 16, 32, 64, 128, 238, 500, 1000):
            A = torch.randn(M, K, device=dev)
            out = torch.empty(M, N, device=dev)
            res = {}
            if M <= 128:
                res["small"] = timeit(lambda: ops.gemm(A, W, bias=b, out=out, small=True))
            res["splitk"] = timeit(lambda: ops.gemm(A, W, bias=b, out=out, splitk=True))
            res["mfma32"] = timeit(lambda: ops.gemm(A, W, bias=b, out=out, splitk=False))
            res["bf16x6"] = timeit(lambda: ops.gemm(A, Ws, bias=b, out=out))
            print("N=%4d K=%4d M=%4d  " % (N, K, M) + "  ".join("%s %6.1f us" % kv for kv in res.items()))


if __name__ == "__main__":
    main()
