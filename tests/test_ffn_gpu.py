"""Fused FFN block (csrc/ffn_fused.hip) against an fp64 torch reference of
`norm(x + linear2(relu(linear1(x))))` (deformable_transformer.py:266-273 / 352-369) and against the three-launch path."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(M, F, seed, wscale=0.05, xscale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((M, 256), generator=g) * xscale
    w1 = torch.randn((F, 256), generator=g) * wscale
    b1 = torch.randn((F,), generator=g) * 0.1
    w2 = torch.randn((256, F), generator=g) * wscale
    b2 = torch.randn((256,), generator=g) * 0.1
    ga = 1.0 + 0.2 * torch.randn((256,), generator=g)
    be = 0.1 * torch.randn((256,), generator=g)
    return x, w1, b1, w2, b2, ga, be


def _ref(x, w1, b1, w2, b2, ga, be):
    d = lambda t: t.double()
    y = d(x) + torch.relu(d(x) @ d(w1).T + d(b1)) @ d(w2).T + d(b2)
    return torch.nn.functional.layer_norm(y, (256,), d(ga), d(be), 1e-5)


@pytest.mark.parametrize("M,F", [(1, 1024), (33, 1024), (128, 1024), (129, 64), (1000, 1024), (2500, 1024), (4097, 96)])
def test_fused_ffn_vs_fp64(M, F):
    from gomatching_amd import ops
    t = _case(M, F, seed=M + F)
    x, w1, b1, w2, b2, ga, be = [v.to(DEV) for v in t]
    ffn = ops.FusedFFN(w1, b1, w2, b2, ga, be)
    y = ops.ffn_fused_ln(x, ffn)
    torch.cuda.synchronize()
    ops.check_range_flag(DEV)
    ref = _ref(*t)
    err = float((y.cpu().double() - ref).abs().max())
    assert err <= 2e-5, err
    # the three-launch path of the same back-end
    old = ops.GEMM_MODE
    ops.GEMM_MODE = "f16x3"
    try:
        h = ops.gemm(x, ops.prep_weight(w1), bias=b1, relu=True)
        z = ops.layernorm(ops.gemm(h, ops.prep_weight(w2), bias=b2, R=x), ga, be)
    finally:
        ops.GEMM_MODE = old
    assert float((y - z).abs().max()) <= 2e-5


@pytest.mark.parametrize("M", [128 * (1024 + 20) - 50, 128 * 1024 + 1, 128 * (1024 + 128), 128 * (1024 + 129) - 7])
def test_fused_ffn_half_height_tail_is_bit_identical(M):
    """Long launches whose last round of one-per-CU tiles is less than half full run that round as half-height (64-row) tiles
    (csrc/ffn_fused.hip, RG = 1): the same bits as the single launch of 128-row tiles, ragged tails included; a last round that
    is more than half full stays one launch."""
    from gomatching_amd import lib, ops
    t = _case(1, 128, seed=3)
    g = torch.Generator().manual_seed(M)
    x = torch.randn((M, 256), generator=g).to(DEV)
    _, w1, b1, w2, b2, ga, be = [v.to(DEV) for v in t]
    ffn = ops.FusedFFN(w1, b1, w2, b2, ga, be)
    L = lib.load()
    try:
        L.gom_ffn_set_half_tail(0)
        one = ops.ffn_fused_ln(x, ffn)
        L.gom_ffn_set_half_tail(1)
        two = ops.ffn_fused_ln(x, ffn)
    finally:
        L.gom_ffn_set_half_tail(1)
    assert torch.equal(one, two)
    ref = _ref(x[-300:].cpu(), *[v.cpu() for v in (w1, b1, w2, b2, ga, be)])
    assert float((two[-300:].cpu().double() - ref).abs().max()) <= 2e-5
    ops.check_range_flag(DEV)


def test_fused_ffn_wide_range_weights_and_strided_rows():
    """Row scales spanning 1e-4..1e2 (trained-like dynamic range), rows of a wider buffer, output in place."""
    from gomatching_amd import ops
    M, F = 777, 1024
    x, w1, b1, w2, b2, ga, be = _case(M, F, seed=5)
    g = torch.Generator().manual_seed(9)
    w1 = w1 * torch.exp(torch.empty((F, 1)).uniform_(-6, 2, generator=g))
    w2 = w2 * torch.exp(torch.empty((256, 1)).uniform_(-6, 2, generator=g))
    ref = _ref(x, w1, b1, w2, b2, ga, be)
    buf = torch.zeros((M, 384), device=DEV)
    buf[:, :256] = x.to(DEV)
    xv = buf[:, :256]
    ffn = ops.FusedFFN(*[v.to(DEV) for v in (w1, b1, w2, b2, ga, be)])
    y = ops.ffn_fused_ln(xv, ffn, out=xv)
    torch.cuda.synchronize()
    assert y.data_ptr() == buf.data_ptr()
    assert float((buf[:, :256].cpu().double() - ref).abs().max()) <= 5e-5
    assert float(buf[:, 256:].abs().max()) == 0.0


def test_fused_ffn_flags_an_activation_beyond_fp16():
    from gomatching_amd import ops, lib
    x, w1, b1, w2, b2, ga, be = [v.to(DEV) for v in _case(64, 1024, seed=1, xscale=1e5)]
    ffn = ops.FusedFFN(w1, b1, w2, b2, ga, be)
    ops.ffn_fused_ln(x, ffn)
    torch.cuda.synchronize()
    with pytest.raises(lib.GomError):
        ops.check_range_flag(DEV)


@pytest.mark.parametrize("M,relu_out", [(1, False), (300, True), (20000, False), (4099, True)])
def test_two_layer_perceptron_form(M, relu_out):
    """The fused FFN kernel as a plain 256 -> 256 -> 256 perceptron (ref_point_head, the first two layers of the coordinate /
    boundary heads) against two launches of the row-resident GEMM and float64."""
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(M)
    dev = "cuda"
    w1 = (torch.randn(256, 256, generator=g) / 16).to(dev)
    b1 = (torch.randn(256, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(256, 256, generator=g) / 16).to(dev)
    b2 = (torch.randn(256, generator=g) * 0.1).to(dev)
    x = torch.randn(M, 256, generator=g).to(dev)
    blk = ops.FusedMLP2(w1, b1, w2, b2, relu_out)
    got = ops.mlp2_fused(x, blk)
    h = ops.gemm(x, ops.split_weight(w1, kind="f16x3"), bias=b1, relu=True)
    two = ops.gemm(h, ops.split_weight(w2, kind="f16x3"), bias=b2, relu=relu_out)
    ops.check_range_flag(torch.device(dev, torch.cuda.current_device()))
    assert float((got - two).abs().max()) < 1e-5
    d = lambda t: t.double().cpu()
    y = torch.relu(d(x) @ d(w1).t() + d(b1)) @ d(w2).t() + d(b2)
    if relu_out:
        y = torch.relu(y)
    assert float((d(got) - y).abs().max()) < 2e-5
