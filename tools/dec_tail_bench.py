"""Interleaved A/B on one GPU at the decoder's shape (M = 8 x 100 x 25 rows): the tail of a decoder layer as ONE launch
(csrc/dec_tail.hip) against the four launches it replaces (fused FFN, two-layer perceptron, ref_update, two-layer perceptron)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gomatching_amd import ops                                   # noqa: E402
from test_dec_tail_gpu import _case                              # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
x, ffn, coord, qpos, ref, dim_t = _case(M, 1024, seed=3)
dv = lambda t: t.to("cuda")
blk = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t), form=1)
blk2 = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t), form=2, waves=4)
blk8 = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t), form=2, waves=8)
f = ops.FusedFFN(*[dv(v) for v in ffn])
c = ops.FusedMLP2(dv(coord[0][0]), dv(coord[0][1]), dv(coord[1][0]), dv(coord[1][1]), True)
q = ops.FusedMLP2(dv(qpos[0][0]), dv(qpos[0][1]), dv(qpos[1][0]), dv(qpos[1][1]), False)
X, R, DT, W3 = dv(x), dv(ref), dv(dim_t), (dv(coord[2][0]), dv(coord[2][1]))


def four():
    y = ops.ffn_fused_ln(X, f)
    r, e = ops.ref_update(ops.mlp2_fused(y, c), W3, R, DT, want_pos=True)
    return y, r, ops.mlp2_fused(e, q)


def one():
    return ops.dec_tail(X, blk, R, want_qpos=True)


def one_last():
    return ops.dec_tail(X, blk, R, want_qpos=False)


g = torch.Generator().manual_seed(1)
samp = dv(torch.randn((M, 256), generator=g))
wo, bo = dv(torch.randn((256, 256), generator=g) / 16), dv(torch.randn((256,), generator=g) * 0.1)
pg, pb = dv(1.0 + 0.2 * torch.randn((256,), generator=g)), dv(0.1 * torch.randn((256,), generator=g))
blk_p = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                    proj_w=(wo, bo, pg, pb), form=1)
blk_p2 = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                     proj_w=(wo, bo, pg, pb), form=2, waves=4)
blk_p8 = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                     proj_w=(wo, bo, pg, pb), form=2, waves=8)
pl = ops.proj_ln_block((ops.prep_weight(wo), bo), (pg, pb))


def five():
    t3 = ops.proj_ln(samp, pl, X)
    y = ops.ffn_fused_ln(t3, f)
    r, e = ops.ref_update(ops.mlp2_fused(y, c), W3, R, DT, want_pos=True)
    return y, r, ops.mlp2_fused(e, q)


def one_proj():
    return ops.dec_tail(samp, blk_p, R, want_qpos=True, residual=X)


def one2():
    return ops.dec_tail(X, blk2, R, want_qpos=True)


def one2_last():
    return ops.dec_tail(X, blk2, R, want_qpos=False)


def one2_proj():
    return ops.dec_tail(samp, blk_p2, R, want_qpos=True, residual=X)


def one8():
    return ops.dec_tail(X, blk8, R, want_qpos=True)


def one8_last():
    return ops.dec_tail(X, blk8, R, want_qpos=False)


def one8_proj():
    return ops.dec_tail(samp, blk_p8, R, want_qpos=True, residual=X)


def ffn_only():
    return ops.ffn_fused_ln(X, f)


def burst(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


a, b = four(), one()
print("max |d| tgt %.2e ref %.2e qpos %.2e" % tuple(float((u - v).abs().max()) for u, v in zip(a, b)))
b2 = one2()
print("form 2 vs form 1: max |d| tgt %.2e ref %.2e qpos %.2e" % tuple(float((u - v).abs().max()) for u, v in zip(b, b2)))
print("form 2 vs form 1 with out_proj: max |d| tgt %.2e ref %.2e qpos %.2e" % tuple(float((u - v).abs().max()) for u, v in zip(one_proj(), one2_proj())))
print("form 2, eight waves vs four: max |d| tgt %.2e ref %.2e qpos %.2e | with out_proj %.2e %.2e %.2e" % (
    tuple(float((u - v).abs().max()) for u, v in zip(b2, one8())) + tuple(float((u - v).abs().max()) for u, v in zip(one2_proj(), one8_proj()))))
for rnd in range(4):
    print("round %d  M = %d: four launches %.1f us | one launch %.1f us | one launch, last layer (no qpos) %.1f us | fused FFN alone %.1f us"
          " || with out_proj + norm_cross: five launches %.1f us | one launch %.1f us"
          % (rnd, M, burst(four), burst(one), burst(one_last), burst(ffn_only), burst(five), burst(one_proj)), flush=True)
    print("         form 2 (csrc/dec_tail2.hip): one launch %.1f us | last layer %.1f us | with out_proj + norm_cross %.1f us"
          % (burst(one2), burst(one2_last), burst(one2_proj)), flush=True)
    print("         form 2, EIGHT waves (two per SIMD): one launch %.1f us | last layer %.1f us | with out_proj + norm_cross %.1f us"
          % (burst(one8), burst(one8_last), burst(one8_proj)), flush=True)
