"""An encoder layer's out_proj + norm1 + FFN + norm2: ONE launch (gom_proj_ffn_ln_f32, csrc/dec_tail.hip FFN_ONLY form) against the
proj_ln launch + the fused FFN launch, alternating bursts on one GPU at the encoder's shape (M = 8 x 37 171 tokens)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops                                   # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 37171
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(s, generator=g).to("cuda")
samp, src = r(M, 256), r(M, 256)
wo, bo, g1, be1 = r(256, 256) / 16, r(256) * 0.1, 1.0 + 0.2 * r(256), 0.1 * r(256)
w1, b1, w2, b2, g2, be2 = r(1024, 256) * 0.05, r(1024) * 0.1, r(256, 1024) * 0.05, r(256) * 0.1, 1.0 + 0.2 * r(256), 0.1 * r(256)
blk = ops.ProjFFN(wo, bo, g1, be1, w1, b1, w2, b2, g2, be2)
pl = ops.proj_ln_block((ops.prep_weight(wo), bo), (g1, be1))
f = ops.FusedFFN(w1, b1, w2, b2, g2, be2)


def burst(fn, n=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


one = lambda: ops.proj_ffn_ln(samp, blk, src)
two = lambda: ops.ffn_fused_ln(ops.proj_ln(samp, pl, src), f)
print("max |d| one launch vs two: %.2e" % float((one() - two()).abs().max()))
for rnd in range(4):
    print("round %d  M = %d: proj_ln + fused FFN %.1f us (proj_ln alone %.1f, FFN alone %.1f) | one launch %.1f us"
          % (rnd, M, burst(two), burst(lambda: ops.proj_ln(samp, pl, src)), burst(lambda: ops.ffn_fused_ln(src, f)), burst(one)), flush=True)
