"""The tracker's skinny products (M <= 64 rows, K = 1024) on an idle GPU: gemm_small (8x8 patch per wave, VALU) against the
deterministic split-K exact-fp32 MFMA path it replaced for M*N > 2^16; bursts of back-to-back launches, alternating."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"


def burst(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


g = torch.Generator().manual_seed(0)
for M, N, K in ((50, 3072, 1024), (22, 3072, 1024), (50, 1024, 1024), (10, 1024, 1024), (10, 2048, 1024), (10, 50, 1024), (64, 1024, 1024)):
    A = torch.randn((M, K), generator=g).to(dev)
    W = torch.randn((N, K), generator=g).to(dev)
    b = torch.randn((N,), generator=g).to(dev)
    fa = lambda: ops.gemm(A, W, bias=b, small=True)
    fb = lambda: ops.gemm(A, W, bias=b, splitk=True)
    for f in (fa, fb):
        f()
    torch.cuda.synchronize()
    ta, tb = [], []
    for _ in range(5):
        ta.append(burst(fa))
        tb.append(burst(fb))
    print("M %3d N %4d K %4d: gemm_small %6.1f us | split-K MFMA %6.1f us | max|d| %.2e" % (
        M, N, K, sorted(ta)[2], sorted(tb)[2], float((fa() - fb()).abs().max())), flush=True)
