"""GPU: the decoder's self-attention blocks as one launch each (csrc/dec_attn.hip) against a float64 statement of
nn.MultiheadAttention + residual + LayerNorm (deformable_transformer.py:386-404), and against the five-launch path."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    in_w = torch.randn(768, 256, generator=g) / 16 * torch.logspace(-1, 1, 768).view(-1, 1) ** 0.3
    in_b = torch.randn(768, generator=g) * 0.1
    out_w = torch.randn(256, 256, generator=g) / 16
    out_b = torch.randn(256, generator=g) * 0.1
    gamma = torch.rand(256, generator=g) + 0.5
    beta = torch.randn(256, generator=g) * 0.1
    return in_w, in_b, out_w, out_b, gamma, beta


def _ref(x, pos, w, groups):
    """groups: LongTensor [n_groups, G] of row indices; float64."""
    in_w, in_b, out_w, out_b, gamma, beta = [t.double() for t in w]
    x = x.double()
    qk_in = x if pos is None else x + pos.double()
    q = qk_in @ in_w[:256].t() + in_b[:256]
    k = qk_in @ in_w[256:512].t() + in_b[256:512]
    v = x @ in_w[512:].t() + in_b[512:]
    out = torch.zeros_like(x)
    n, G = groups.shape
    qg = q[groups].view(n, G, 8, 32).transpose(1, 2)
    kg = k[groups].view(n, G, 8, 32).transpose(1, 2)
    vg = v[groups].view(n, G, 8, 32).transpose(1, 2)
    a = torch.softmax(qg @ kg.transpose(-1, -2) / math.sqrt(32.0), -1)
    o = (a @ vg).transpose(1, 2).reshape(n * G, 256)
    y = o @ out_w.t() + out_b + x[groups.reshape(-1)]
    y = torch.nn.functional.layer_norm(y, (256,), gamma, beta, 1e-5)
    out[groups.reshape(-1)] = y
    return out


FORMS = [1, 2]                                                   # csrc/dec_attn.hip | csrc/dec_attn2.hip (16-token waves, two per SIMD)


def _block(ops, w, inter, form=None):
    d = [t.to(DEV) for t in w]
    return ops.DecAttnBlock(d[0], d[1], d[2], d[3], d[4], d[5], inter, form=form)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("groups,G", [(1, 25), (4, 25), (37, 25), (800, 25), (5, 32), (9, 3), (6, 1), (3, 16), (7, 17)])
def test_intra_block(groups, G, form):
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(groups * 31 + G)
    w = _weights(1)
    rows = groups * G
    x, pos = torch.randn(rows, 256, generator=g), torch.randn(rows, 256, generator=g) * 0.7
    blk = _block(ops, w, False, form)
    out = torch.full((rows + 3, 256), 7.0, device=DEV)
    xd = torch.cat([x, torch.zeros(3, 256)]).to(DEV)
    pd = torch.cat([pos, torch.zeros(3, 256)]).to(DEV)
    ops.dec_attn(xd, blk, groups, G, pos=pd, out=out)
    ref = _ref(x, pos, w, torch.arange(rows).view(groups, G))
    err = float((out[:rows].cpu().double() - ref).abs().max())
    assert err < 2e-5, err
    assert float((out[rows:] - 7.0).abs().max()) == 0.0             # rows beyond the groups are not touched
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("B,nq,P", [(1, 100, 25), (2, 12, 25), (8, 100, 25), (1, 128, 3), (3, 7, 2), (2, 1, 5), (1, 97, 1), (1, 121, 2)])
def test_inter_block(B, nq, P, form):
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(B * 131 + nq * 7 + P)
    w = _weights(2)
    rows = B * nq * P
    x = torch.randn(rows, 256, generator=g)
    blk = _block(ops, w, True, form)
    out = torch.full((rows, 256), 7.0, device=DEV)
    ops.dec_attn(x.to(DEV), blk, B * P, nq, inner=P, out=out)
    idx = torch.arange(rows).view(B, nq, P).permute(0, 2, 1).reshape(B * P, nq)
    ref = _ref(x, None, w, idx)
    err = float((out.cpu().double() - ref).abs().max())
    assert err < 2e-5, err
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


@pytest.mark.parametrize("B,nq,P", [(8, 300, 25), (1, 300, 25), (2, 129, 3), (1, 352, 1), (1, 161, 2), (1, 320, 2)])
def test_inter_block_more_than_128_queries(B, nq, P):
    """GoMatching++ (300 queries): in_proj + attention per (group, head) (csrc/dec_inter.hip) + the out_proj / LayerNorm launch
    against the float64 statement of the whole block, and the head outputs alone against float64 too."""
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(B * 131 + nq * 7 + P)
    w = _weights(2)
    rows = B * nq * P
    x = torch.randn(rows, 256, generator=g)
    blk = _block(ops, w, True)
    xd = x.to(DEV)
    heads = torch.full((rows, 256), 7.0, device=DEV)
    ops.dec_inter_heads(xd, blk, B * P, nq, inner=P, out=heads)
    idx = torch.arange(rows).view(B, nq, P).permute(0, 2, 1).reshape(B * P, nq)
    in_w, in_b = w[0].double(), w[1].double()
    q, k, v = [(x.double() @ in_w[i * 256:(i + 1) * 256].t() + in_b[i * 256:(i + 1) * 256])[idx].view(B * P, nq, 8, 32).transpose(1, 2)
               for i in range(3)]
    o = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32.0), -1) @ v).transpose(1, 2).reshape(B * P * nq, 256)
    ref_heads = torch.zeros(rows, 256, dtype=torch.float64)
    ref_heads[idx.reshape(-1)] = o
    err = float((heads.cpu().double() - ref_heads).abs().max())
    assert err < 1e-5, err
    d = [t.to(DEV) for t in w]
    pl = ops.ProjLN(ops.split_weight(d[2], kind="f16x3"), d[3], d[4], d[5])
    out = ops.proj_ln(heads, pl, xd)
    err = float((out.cpu().double() - _ref(x, None, w, idx)).abs().max())
    assert err < 2e-5, err
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))
    if nq == 300 and B == 1:                                         # the range contract of this kernel
        xb = xd.clone()
        xb[37, 5] = 7e4
        ops.dec_inter_heads(xb, blk, B * P, nq, inner=P)
        with pytest.raises(Exception, match="fp16's range"):
            ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


@pytest.mark.parametrize("form", FORMS)
def test_blocks_raise_the_range_flag(form):
    from gomatching_amd import ops
    dev = torch.device(DEV, torch.cuda.current_device())
    w = _weights(3)
    ops.check_range_flag(dev)
    x = torch.randn(100, 256, device=DEV)
    x[37, 5] = 7e4
    ops.dec_attn(x, _block(ops, w, False, form), 4, 25, pos=torch.zeros_like(x))
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(dev)
    ops.dec_attn(x, _block(ops, w, True, form), 4, 25, inner=1)
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(dev)
    big = list(w)
    big[0] = w[0].clone()
    big[0][300] *= 1e6                                               # one k feature beyond fp16 after the projection
    ops.dec_attn(torch.randn(100, 256, device=DEV), _block(ops, big, True, form), 4, 25, inner=1)
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(dev)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("B,nq,P", [(1, 100, 25), (8, 100, 25), (2, 12, 25), (1, 128, 3), (3, 7, 2), (1, 97, 1)])
def test_inter_block_with_offsets_and_logits_behind_it(B, nq, P, form):
    """The RAW form: the inter block's launch also makes the cross attention's sampling_offsets | attention_weights product,
    raw = (out + query_pos) Wraw^T + braw (ms_deform_attn.py:117-131 on query = tgt + query_pos).  `out` must be the plain
    form's bits; raw against float64 and against the row-resident GEMM launch it replaces."""
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(B * 131 + nq * 7 + P + 1)
    w = _weights(2)
    rows = B * nq * P
    x, qpos = torch.randn(rows, 256, generator=g), torch.randn(rows, 256, generator=g) * 0.7
    rw = torch.randn(384, 256, generator=g) / 16 * torch.logspace(-1, 1, 384).view(-1, 1) ** 0.3
    rb = torch.randn(384, generator=g) * 0.1
    d = [t.to(DEV) for t in w]
    blk = ops.DecAttnBlock(d[0], d[1], d[2], d[3], d[4], d[5], True, raw=(rw.to(DEV), rb.to(DEV)), form=form)
    plain = _block(ops, w, True, form)
    xd, qd = x.to(DEV), qpos.to(DEV)
    out, raw = ops.dec_attn(xd, blk, B * P, nq, inner=P, raw_pos=qd)
    want = ops.dec_attn(xd, plain, B * P, nq, inner=P)
    assert torch.equal(out, want)
    idx = torch.arange(rows).view(B, nq, P).permute(0, 2, 1).reshape(B * P, nq)
    ref = _ref(x, None, w, idx)
    ref_raw = (ref + qpos.double()) @ rw.double().t() + rb.double()
    err = float((raw.cpu().double() - ref_raw).abs().max())
    assert err < 5e-5, err
    lin = ops.k256_linear(ops.split_weight(rw.to(DEV), kind="f16x3"), rb.to(DEV))
    two = ops.linear(want, lin, A2=qd)
    assert float((raw - two).abs().max()) < 2e-5
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


@pytest.mark.parametrize("form", FORMS)
def test_blocks_do_not_depend_on_what_shares_the_launch(form):
    """Batch invariance (a tracker property: the same query gives the same bits whatever else is in the clip): a group's output is
    the same in a launch of many groups as in a launch of its own -- both forms, both blocks, the RAW form's second output too."""
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(17)
    w = _weights(4)
    # intra: 37 groups of 25 points; groups 8 .. 11 alone (a whole workgroup of the eight-wave form) and group 36 alone (a tail workgroup)
    x, pos = torch.randn(37 * 25, 256, generator=g).to(DEV), (torch.randn(37 * 25, 256, generator=g) * 0.7).to(DEV)
    blk = _block(ops, w, False, form)
    full = ops.dec_attn(x, blk, 37, 25, pos=pos)
    for g0, n in ((8, 4), (36, 1), (5, 1)):
        sub = ops.dec_attn(x[g0 * 25:(g0 + n) * 25].contiguous(), blk, n, 25, pos=pos[g0 * 25:(g0 + n) * 25].contiguous())
        assert torch.equal(sub, full[g0 * 25:(g0 + n) * 25]), (g0, n)
    # inter (+ offsets | logits): 3 frames x 100 queries x 5 points; frame 1 alone
    B, nq, P = 3, 100, 5
    x, qpos = torch.randn(B * nq * P, 256, generator=g).to(DEV), (torch.randn(B * nq * P, 256, generator=g) * 0.7).to(DEV)
    rw = (torch.randn(384, 256, generator=g) / 16).to(DEV)
    rb = (torch.randn(384, generator=g) * 0.1).to(DEV)
    d = [t.to(DEV) for t in w]
    blk = ops.DecAttnBlock(d[0], d[1], d[2], d[3], d[4], d[5], True, raw=(rw, rb), form=form)
    out, raw = ops.dec_attn(x, blk, B * P, nq, inner=P, raw_pos=qpos)
    one = slice(nq * P, 2 * nq * P)
    o1, r1 = ops.dec_attn(x[one].contiguous(), blk, P, nq, inner=P, raw_pos=qpos[one].contiguous())
    assert torch.equal(o1, out[one]) and torch.equal(r1, raw[one])
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))
