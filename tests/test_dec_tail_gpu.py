"""The row-local tail of a composite decoder layer as one launch -- form 2: the CU-cooperative kernel of csrc/dec_tail2.hip (80 rows
per workgroup, the waves split the output columns; round 6, the default), form 1: csrc/dec_tail.hip -- against a float64 statement of
deformable_transformer.py:352-369 (FFN + norm3), :484-488 (ctrl_point_coord + reference refinement) and :470-473 +
adet/modeling/model/utils.py:24-37 (the next layer's ref_point_head over the sine embedding), and against the four-launch path
it replaces (fused FFN, two-layer perceptron, ref_update, two-layer perceptron)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(M, F, seed, xscale=1.0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(s, generator=g)
    x = r(M, 256) * xscale
    ffn = (r(F, 256) * 0.05, r(F) * 0.1, r(256, F) * 0.05, r(256) * 0.1, 1.0 + 0.2 * r(256), 0.1 * r(256))
    coord = [(r(256, 256) / 16, r(256) * 0.1), (r(256, 256) / 16, r(256) * 0.1), (r(2, 256) / 16, r(2) * 0.1)]
    qpos = [(r(256, 256) / 16, r(256) * 0.1), (r(256, 256) / 16, r(256) * 0.1)]
    ref = torch.rand((M, 2), generator=g)
    ref[: min(M, 4)] = torch.tensor([[0.0, 1.0], [1.0, 0.0], [1e-7, 0.5], [0.999999, 0.3]])[: min(M, 4)]   # the clamps of inverse_sigmoid
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / 128)
    return x, ffn, coord, qpos, ref, dim_t


def _ref64(x, ffn, coord, qpos, ref, dim_t):
    d = lambda t: t.double()
    w1, b1, w2, b2, ga, be = ffn
    y = d(x) + torch.relu(d(x) @ d(w1).T + d(b1)) @ d(w2).T + d(b2)
    y = torch.nn.functional.layer_norm(y, (256,), d(ga), d(be), 1e-5)
    h = y
    for i, (w, b) in enumerate(coord):
        h = h @ d(w).T + d(b)
        if i < 2:
            h = torch.relu(h)
    r = d(ref).clamp(0, 1)
    inv = torch.log(r.clamp(min=1e-5) / (1 - r).clamp(min=1e-5))
    new_ref = torch.sigmoid(h + inv)
    # gen_point_pos_embed: pos = pts * 2 pi / dim_t; stack(sin(even), cos(odd)); cat(x-half, y-half)
    pos = new_ref[:, :, None] * (2 * math.pi) / d(dim_t)[None, None, :]
    emb = torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2)          # [M, 2, 128]
    emb = torch.cat((emb[:, 0], emb[:, 1]), dim=-1)
    q = torch.relu(emb @ d(qpos[0][0]).T + d(qpos[0][1])) @ d(qpos[1][0]).T + d(qpos[1][1])
    return y, new_ref, q


def _fw(code):
    """test parameter -> DecTail keywords: 1 = csrc/dec_tail.hip, 2 = csrc/dec_tail2.hip with four waves, 28 = with eight (two per SIMD)"""
    return {"form": 1} if code == 1 else {"form": 2, "waves": 8 if code == 28 else 4}


@pytest.mark.parametrize("form", [28, 2, 1])
@pytest.mark.parametrize("M,F,want", [(1, 1024, True), (33, 1024, True), (128, 64, False), (129, 1024, True), (2500, 1024, False),
                                       (20000, 1024, True), (4097, 96, True), (79, 128, True), (80, 256, False), (161, 1024, True)])
def test_dec_tail_vs_fp64_and_four_launches(M, F, want, form):
    from gomatching_amd import ops
    if form != 1 and F % 128:
        pytest.skip("form 2 takes the hidden layer in chunks of 128 (ops.DecTail falls back to form 1)")
    x, ffn, coord, qpos, ref, dim_t = _case(M, F, seed=M + F)
    dv = lambda t: t.to(DEV)
    blk = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                      **_fw(form))
    assert blk.form == _fw(form)["form"]
    y, nref, qp = ops.dec_tail(dv(x), blk, dv(ref), want_qpos=want)
    torch.cuda.synchronize()
    ops.check_range_flag(DEV)
    assert (qp is None) == (not want)
    ry, rref, rq = _ref64(x, ffn, coord, qpos, ref, dim_t)
    assert float((y.cpu().double() - ry).abs().max()) <= 2e-5
    assert float((nref.cpu().double() - rref).abs().max()) <= 2e-6
    if want:
        assert float((qp.cpu().double() - rq).abs().max()) <= 3e-5
    # the path it replaces, same back-end: fused FFN -> two-layer perceptron -> ref_update -> two-layer perceptron
    f = ops.FusedFFN(*[dv(v) for v in ffn])
    c = ops.FusedMLP2(dv(coord[0][0]), dv(coord[0][1]), dv(coord[1][0]), dv(coord[1][1]), True)
    q = ops.FusedMLP2(dv(qpos[0][0]), dv(qpos[0][1]), dv(qpos[1][0]), dv(qpos[1][1]), False)
    y4 = ops.ffn_fused_ln(dv(x), f)
    nref4, emb4 = ops.ref_update(ops.mlp2_fused(y4, c), (dv(coord[2][0]), dv(coord[2][1])), dv(ref), dv(dim_t), want_pos=want)
    assert float((y - y4).abs().max()) <= 2e-5
    assert float((nref - nref4).abs().max()) <= 2e-6
    if want:
        assert float((qp - ops.mlp2_fused(emb4, q)).abs().max()) <= 3e-5


@pytest.mark.parametrize("proj", [False, True])
@pytest.mark.parametrize("form", [28, 2, 1])
def test_dec_tail_rows_are_independent_of_the_launch(form, proj):
    """Batch invariance: a row's bits do not depend on what shares its launch (tile position, tail tile, launch length) -- with and
    without the out_proj block in front (whose residual the form-1 kernel parks in Y: ADVICE r5)."""
    from gomatching_amd import ops
    M = 1000
    x, ffn, coord, qpos, ref, dim_t = _case(M, 1024, seed=7)
    g = torch.Generator().manual_seed(5)
    samp = torch.randn((M, 256), generator=g)
    pw = (torch.randn((256, 256), generator=g) / 16, torch.randn((256,), generator=g) * 0.1, 1.0 + 0.2 * torch.randn((256,), generator=g),
          0.1 * torch.randn((256,), generator=g))
    dv = lambda t: t.to(DEV)
    blk = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                      proj_w=tuple(dv(v) for v in pw) if proj else None, **_fw(form))
    run = (lambda a, b: ops.dec_tail(dv(samp[a:b]).contiguous(), blk, dv(ref[a:b]).contiguous(), residual=dv(x[a:b]).contiguous())) if proj \
        else (lambda a, b: ops.dec_tail(dv(x[a:b]).contiguous(), blk, dv(ref[a:b]).contiguous()))
    full = run(0, M)
    for a, b in ((0, 1), (17, 300), (511, 1000), (900, 901), (80, 160), (79, 241)):
        part = run(a, b)
        for u, v in zip(full, part):
            assert torch.equal(u[a:b], v)


@pytest.mark.parametrize("form", [28, 2, 1])
def test_dec_tail_flags_an_activation_beyond_fp16(form):
    from gomatching_amd import lib, ops
    x, ffn, coord, qpos, ref, dim_t = _case(64, 1024, seed=1, xscale=1e5)
    dv = lambda t: t.to(DEV)
    blk = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                      **_fw(form))
    ops.dec_tail(dv(x), blk, dv(ref))
    torch.cuda.synchronize()
    with pytest.raises(lib.GomError):
        ops.check_range_flag(DEV)


@pytest.mark.parametrize("form", [28, 2, 1])
@pytest.mark.parametrize("M,want", [(1, True), (130, False), (2500, True), (20000, True)])
def test_dec_tail_with_out_proj_in_front(M, want, form):
    """The launch that also takes the cross attention's out_proj + residual + norm_cross (deformable_transformer.py:406-422):
    float64 reference and the proj_ln launch + the plain tail launch it replaces."""
    from gomatching_amd import ops
    x, ffn, coord, qpos, ref, dim_t = _case(M, 1024, seed=M + 11)
    g = torch.Generator().manual_seed(M)
    samp = torch.randn((M, 256), generator=g)
    wo, bo = torch.randn((256, 256), generator=g) / 16, torch.randn((256,), generator=g) * 0.1
    pg, pb = 1.0 + 0.2 * torch.randn((256,), generator=g), 0.1 * torch.randn((256,), generator=g)
    dv = lambda t: t.to(DEV)
    blk = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                      proj_w=(dv(wo), dv(bo), dv(pg), dv(pb)), **_fw(form))
    y, nref, qp = ops.dec_tail(dv(samp), blk, dv(ref), want_qpos=want, residual=dv(x))
    torch.cuda.synchronize()
    ops.check_range_flag(DEV)
    d = lambda t: t.double()
    t3 = torch.nn.functional.layer_norm(d(x) + d(samp) @ d(wo).T + d(bo), (256,), d(pg), d(pb), 1e-5)
    ry, rref, rq = _ref64(t3, ffn, coord, qpos, ref, dim_t)
    assert float((y.cpu().double() - ry).abs().max()) <= 3e-5
    assert float((nref.cpu().double() - rref).abs().max()) <= 2e-6
    if want:
        assert float((qp.cpu().double() - rq).abs().max()) <= 3e-5
    plain = ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                        form=1)
    pl = ops.proj_ln_block((ops.prep_weight(dv(wo)), dv(bo)), (dv(pg), dv(pb)))
    t3g = ops.proj_ln(dv(samp), pl, dv(x))
    y2, nref2, qp2 = ops.dec_tail(t3g, plain, dv(ref), want_qpos=want)
    assert float((y - y2).abs().max()) <= 3e-5 and float((nref - nref2).abs().max()) <= 2e-6
    if want:
        assert float((qp - qp2).abs().max()) <= 3e-5

