"""Micro-reproducer for the tracker determinism issue: the tracker's small GEMM on a high-priority stream while a heavy GEMM
runs on the default stream; every output is compared with the idle-GPU result.
    python tools/race_repro.py [load kinds ...]      load in {none, bf16x6, f16x3, fp32}"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
loads = sys.argv[1:] or ["none", "bf16x6", "f16x3", "fp32"]
g = torch.Generator().manual_seed(3)
M, N, K = int(os.environ.get("RM", 48)), 1024, 1024
x = torch.randn((M, K), generator=g).to(dev)
w = (torch.randn((N, K), generator=g) * 0.03).to(dev)
b = torch.randn((N,), generator=g).to(dev)
r = torch.randn((M, N), generator=g).to(dev)
ITER = int(os.environ.get("ITER", 3000))
variant = os.environ.get("VARIANT", "small")


def small(out):
    if variant == "small":
        return ops.gemm(x, w, bias=b, R=r, small=True, out=out)
    if variant == "splitk":
        return ops.gemm(x, w, bias=b, R=r, splitk=True, out=out)
    return ops.gemm(x, w, bias=b, R=r, out=out)


ref = small(torch.empty((M, N), device=dev))
torch.cuda.synchronize()
A = torch.randn((16384, 1024), generator=g).to(dev)
W = (torch.randn((1024, 1024), generator=g) * 0.03).to(dev)
trk = torch.cuda.Stream(priority=-1)
for load in loads:
    Wl = None
    if load in ("bf16x6", "f16x3"):
        Wl = ops.split_weight(W, kind=load)
    elif load == "fp32":
        Wl = W
    big_out = torch.empty((A.shape[0], 1024), device=dev)
    outs = [torch.empty((M, N), device=dev) for _ in range(8)]
    bad = torch.zeros((1,), dtype=torch.int64, device=dev)
    badrows = torch.zeros((M,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for it in range(ITER):
        if Wl is not None and it % 4 == 0:
            ops.gemm(A, Wl, out=big_out)                    # default stream: keeps the chip busy
        with torch.cuda.stream(trk):
            o = small(outs[it % 8])
            ne = o != ref
            bad += ne.any().long()
            badrows += ne.any(1).long()
    torch.cuda.synchronize()
    print("load %-7s variant %s: %d of %d launches differ from the idle result; rows hit %s" % (
        load, variant, int(bad), ITER, torch.nonzero(badrows).flatten().tolist()), flush=True)
