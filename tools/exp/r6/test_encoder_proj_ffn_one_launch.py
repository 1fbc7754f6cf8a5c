"""(removed from tests/test_dec_tail_gpu.py with the experiment: tools/exp/r6/encoder_proj_ffn_one_launch.patch)"""
import pytest
import torch

DEV = "cuda"


@pytest.mark.parametrize("M", [1, 127, 128, 129, 5000, 37171])
def test_encoder_proj_ffn_one_launch(M):
    """gom_proj_ffn_ln_f32 (round 6): an encoder layer's out_proj + norm1 + FFN + norm2 as one launch (csrc/dec_tail.hip's first two
    blocks; deformable_transformer.py:258-278) against a float64 statement and against the two launches it replaces."""
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(M)
    r = lambda *s: torch.randn(s, generator=g)
    samp, src = r(M, 256), r(M, 256)
    wo, bo, g1, be1 = r(256, 256) / 16, r(256) * 0.1, 1.0 + 0.2 * r(256), 0.1 * r(256)
    w1, b1, w2, b2, g2, be2 = r(1024, 256) * 0.05, r(1024) * 0.1, r(256, 1024) * 0.05, r(256) * 0.1, 1.0 + 0.2 * r(256), 0.1 * r(256)
    dv = lambda t: t.to(DEV)
    blk = ops.ProjFFN(dv(wo), dv(bo), dv(g1), dv(be1), dv(w1), dv(b1), dv(w2), dv(b2), dv(g2), dv(be2))
    y = ops.proj_ffn_ln(dv(samp), blk, dv(src))
    torch.cuda.synchronize()
    ops.check_range_flag(DEV)
    d = lambda t: t.double()
    t1 = torch.nn.functional.layer_norm(d(src) + d(samp) @ d(wo).T + d(bo), (256,), d(g1), d(be1), 1e-5)
    ref = torch.nn.functional.layer_norm(t1 + torch.relu(t1 @ d(w1).T + d(b1)) @ d(w2).T + d(b2), (256,), d(g2), d(be2), 1e-5)
    assert float((y.cpu().double() - ref).abs().max()) <= 3e-5
    pl = ops.proj_ln_block((ops.prep_weight(dv(wo)), dv(bo)), (dv(g1), dv(be1)))
    f = ops.FusedFFN(dv(w1), dv(b1), dv(w2), dv(b2), dv(g2), dv(be2))
    y2 = ops.ffn_fused_ln(ops.proj_ln(dv(samp), pl, dv(src)), f)
    assert float((y - y2).abs().max()) <= 3e-5
    if M > 200:                                                      # a row's bits do not depend on the launch
        part = ops.proj_ffn_ln(dv(samp[100:M - 50]).contiguous(), blk, dv(src[100:M - 50]).contiguous())
        assert torch.equal(part, y[100:M - 50])
