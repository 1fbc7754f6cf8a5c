"""Can the decoder's idle CUs be used?  The decoder's launches (157-200 one-per-CU workgroups of 512-register waves on 256 CUs) on
one stream, backbone-like HBM-bound launches (res2's fused bottleneck pairs) on another: each alone, then side by side.  If the
pair takes clearly less than the sum, a staggered second detector lane (the next step's backbone under this step's decoder) could
pay; if the decoder's launches stretch by what the backbone gains, it cannot."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gomatching_amd import ops                                   # noqa: E402
from test_dec_tail_gpu import _case as _tail_case                # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
Q = 20000
dv = lambda t: t.to(dev)
x, ffn_w, coord, qpos_w, ref, dim_t = _tail_case(Q, 1024, seed=3)
wo, bo = dv(torch.randn((256, 256), generator=g) / 16), dv(torch.randn((256,), generator=g) * 0.1)
ones, zeros = torch.ones((256,), device=dev), torch.zeros((256,), device=dev)
tail = ops.DecTail(tuple(dv(v) for v in ffn_w), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos_w], dv(dim_t),
                   proj_w=(wo, bo, ones, zeros))
in_w, in_b = dv(torch.randn((768, 256), generator=g) / 16), dv(torch.randn((768,), generator=g) * 0.1)
rw, rb = dv(torch.randn((384, 256), generator=g) / 16), dv(torch.randn((384,), generator=g) * 0.1)
intra = ops.DecAttnBlock(in_w, in_b, wo, bo, ones, zeros, False)
inter = ops.DecAttnBlock(in_w, in_b, wo, bo, ones, zeros, True, raw=(rw, rb))
X, R, pos, samp = dv(x), dv(ref), dv(torch.randn((Q, 256), generator=g)), dv(torch.randn((Q, 256), generator=g))


def decoder_like():
    for _ in range(6):
        t = ops.dec_attn(X, intra, 800, 25, pos=pos)
        t, raw = ops.dec_attn(t, inter, 200, 100, inner=25, raw_pos=pos)
        ops.dec_tail(samp, tail, R, want_qpos=True, residual=t)


k1, mp, hw = 64, 64, (250, 445)
c4 = 4 * k1
a = dv(torch.randn(8, hw[0], hw[1], k1, generator=g).abs())
Rr = dv(torch.randn(8, hw[0], hw[1], c4, generator=g))
s3 = ops.split_weight(dv(torch.randn(c4, k1, generator=g) / k1 ** 0.5), conv_shape=(c4, 1, 1, k1), kind="f16x3")
s1 = ops.split_weight(dv(torch.randn(mp, c4, generator=g) / c4 ** 0.5), conv_shape=(mp, 1, 1, c4), kind="f16x3")
blk = ops.BneckFused(s3, torch.ones(c4, device=dev), torch.zeros(c4, device=dev), s1, torch.ones(mp, device=dev), torch.zeros(mp, device=dev))


def backbone_like():
    for _ in range(5):
        ops.bneck_fused(a, blk, Rr)


def timed(fns, n=6):
    streams = [torch.cuda.Stream() for _ in fns]
    for f, s in zip(fns, streams):
        with torch.cuda.stream(s):
            f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in fns]
    e0.record()
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for f, s, e in zip(fns, streams, ends):
        with torch.cuda.stream(s):
            for _ in range(n):
                f()
            e.record(s)
    torch.cuda.synchronize()
    return [e0.elapsed_time(e) / n for e in ends]


for rnd in range(3):
    ta, = timed([decoder_like])
    tb, = timed([backbone_like])
    tab = timed([decoder_like, backbone_like])
    print("round %d: decoder-like alone %.2f ms | backbone-like alone %.2f ms | side by side: decoder-like done after %.2f ms, backbone-like "
          "after %.2f ms per iteration (sum alone %.2f, max side by side %.2f)" % (rnd, ta, tb, tab[0], tab[1], ta + tb, max(tab)), flush=True)
