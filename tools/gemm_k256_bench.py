"""Row-resident K = 256 kernel (gemm_k256.hip) vs the 128x128 tile kernel on the decoder's Q-side shapes (M = 20 000) and on the
encoder's (M = 297 368): bursts of back-to-back launches, the two kernels ALTERNATING (clock / cache state drifts by more than
the differences looked for), median of the bursts."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

ops.GEMM_MODE = "f16x3"
ops.K256_MAX_ROWS = 1 << 30          # measure the row-resident kernel at every M (ops.linear falls back beyond the default)
dev = "cuda"


def burst(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def ab(fns, rounds=7):
    for f in fns:
        f()
    torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for i, f in enumerate(fns):
            ts[i].append(burst(f))
    return [sorted(t)[len(t) // 2] for t in ts]


g = torch.Generator().manual_seed(0)
for M in (20000, 297368):
    A = torch.randn((M, 256), generator=g).to(dev)
    A2 = torch.randn((M, 256), generator=g).to(dev)
    for N, rc in ((256, 256), (256, 0), (384, 0), (512, 0), (768, 0), (1024, 0)) if M == 20000 else ((256, 256), (256, 0), (640, 384), (1536, 0)):
        W = torch.randn((N, 256), generator=g).to(dev)
        b = torch.randn((N,), generator=g).to(dev)
        R = torch.randn((M, N), generator=g).to(dev) if rc else None
        sw = ops.split_weight(W, kind="f16x3")
        lin = ops.K256Linear(sw, b)
        out = torch.empty((M, N), device=dev)
        fl = 2.0 * M * N * 256
        kw = {"R": R, "r_cols": rc} if rc else {}
        grps = [g_ for g_ in (1, 2, 4) if g_ <= N // 32]
        t = ab([lambda: ops.gemm(A, sw, bias=b, out=out, **kw)] + [(lambda g_: (lambda: ops.linear(A, lin, out=out, groups=g_, **kw)))(g_) for g_ in grps])
        line = "M %6d N %4d R-cols %3d  tile %7.1f us (%5.1f TF) |" % (M, N, rc, t[0], fl / t[0] / 1e6)
        for g_, tt in zip(grps, t[1:]):
            line += " g%d %6.1f us (%5.1f TF)" % (g_, tt, fl / tt / 1e6)
        t2 = ab([lambda: ops.gemm(A, sw, bias=b, A2=A2, out=out), lambda: ops.linear(A, lin, A2=A2, out=out, groups=1)])
        line += " | +A2: add+tile %6.1f, k256 g1 %6.1f" % (t2[0], t2[1])
        print(line, flush=True)

# Long problems: the kernel's two store forms (gom_gemm_k256_set_lines) against the tile kernel, with the encoder's periodic
# position table on the first 384 columns (period = tokens per frame) where r_cols > 0.
from gomatching_amd import lib
M, S = 297368, 37171
A = torch.randn((M, 256), generator=g).to(dev)
for N, rc in ((640, 384), (640, 0), (256, 0), (1536, 0)):
    W = torch.randn((N, 256), generator=g).to(dev)
    b = torch.randn((N,), generator=g).to(dev)
    R = torch.randn((S, rc), generator=g).to(dev) if rc else None
    sw = ops.split_weight(W, kind="f16x3")
    lin = ops.K256Linear(sw, b)
    out = torch.empty((M, N), device=dev)
    kw = {"R": R, "r_cols": rc, "r_period": S} if rc else {}
    ref = ops.gemm(A, sw, bias=b, **kw)
    same = []

    def k256(mode):
        def f():
            lib.load().gom_gemm_k256_set_lines(mode)
            ops.linear(A, lin, out=out, groups=1, **kw)
            lib.load().gom_gemm_k256_set_lines(-1)
        return f
    for mode in (0, 1):
        k256(mode)()
        same.append(bool(torch.equal(out, ref)))
    def plain_order():
        lib.load().gom_gemm_k256_set_interleave(0)
        ops.linear(A, lin, out=out, groups=1, **kw)
        lib.load().gom_gemm_k256_set_interleave(1)
    t = ab([lambda: ops.gemm(A, sw, bias=b, out=out, **kw), k256(0), k256(1), plain_order])
    print("M %6d N %4d periodic R-cols %3d  tile %7.1f us | k256 16-byte stores %7.1f us | k256 whole-line stores %7.1f us | same, tiles in index "
          "order (no frame interleave) %7.1f us | bits equal to the tile kernel: %s" % (M, N, rc, t[0], t[1], t[2], t[3], same), flush=True)
