"""The ResNet stem as one launch (csrc/stem_pool.hip: conv 7x7 / 2 + folded BN + ReLU + max-pool 3x3 / 2) against the two
launches it replaces (gom_conv2d_nhwc_f32_f16x3 + gom_maxpool3x3s2_nhwc_f32; detectron2 BasicStem as built by
/root/reference's adet backbone): the same MFMA sequence per convolution output and an exact maximum, so
the results must be IDENTICAL -- on ragged sizes (patch and image borders everywhere), a one-pixel-high pooled row, the BASELINE
frame size, and through the backbone with the switch on and off."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from gomatching_amd import ops
    ops.GEMM_MODE = "f16x3"
    return ops


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 75, 101), (3, 9, 11), (1, 7, 300), (1, 130, 34), (1, 1000, 1778)])
def test_stem_conv_pool_equals_conv_then_pool(B, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.randn((B, H, W, 4), generator=g).to(DEV) * 1.5
    x[..., 3] = 0.0                                          # the padded fourth channel
    w = ops.prep_conv_weight((torch.randn((64, 7, 7, 4), generator=g) * torch.logspace(-2, 0.5, 64).view(-1, 1, 1, 1)).to(DEV))
    sc = (torch.rand((64,), generator=g) + 0.5).to(DEV)
    sh = torch.randn((64,), generator=g).to(DEV)
    for kw in ({"scale": sc, "shift": sh}, {}):
        ref = ops.maxpool3x3s2(ops.conv2d_nhwc(x, w, relu=True, stride=2, pad=3, **kw))
        got = ops.stem_conv_pool(x, w, **kw)
        assert got.shape == ref.shape
        assert torch.equal(got, ref), float((got - ref).abs().max())
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


def test_stem_range_flag_and_backbone_switch():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    w = ops.prep_conv_weight((torch.randn((64, 7, 7, 4), generator=g) * 0.1).to(DEV))
    dev = torch.device(DEV, torch.cuda.current_device())
    ops.check_range_flag(dev)
    ops.stem_conv_pool(torch.full((1, 32, 32, 4), 7e4, device=DEV), w)      # beyond fp16: flagged, never silent
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(dev)
    # the backbone with the fused stem equals the backbone without it
    from gomatching_amd.modeling.backbone import ResNet50
    from gomatching_amd.weights import synth_state_dict
    from helpers import mini_cfg
    sd = synth_state_dict(mini_cfg(), seed=0)
    x = torch.randn((1, 96, 160, 4), generator=g).to(DEV)
    x[..., 3] = 0.0
    outs = []
    before = ops.STEM_POOL
    try:
        for on in (True, False):
            ops.STEM_POOL = on
            outs.append(ResNet50(sd, DEV).forward(x))
    finally:
        ops.STEM_POOL = before
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
