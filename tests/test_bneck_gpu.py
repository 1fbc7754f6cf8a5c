"""GPU: conv3 + BN + residual + ReLU of a bottleneck block fused with the next block's conv1 + BN + ReLU (csrc/bneck_fused.hip)
against the two launches of the tile kernel it replaces and a float64 statement."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("k1,mp,hw", [(64, 64, (13, 21)), (64, 128, (9, 30)), (128, 128, (12, 11)), (128, 256, (7, 19)),
                                      (64, 64, (125, 223)),
                                      # res4 (256 -> 1024 -> 256): csrc/bneck2.hip, 64-pixel tiles at two workgroups per CU
                                      (256, 256, (1, 1)), (256, 256, (7, 19)), (256, 256, (63, 112)), (128, 256, (125, 223))])
def test_fused_pair_equals_two_launches(k1, mp, hw):
    from gomatching_amd import ops
    g = torch.Generator().manual_seed(k1 + mp + hw[0])
    B, (H, W), c4 = 2, hw, 4 * k1
    a = torch.randn(B, H, W, k1, generator=g).abs().to(DEV)                  # conv2's output is behind a ReLU
    R = torch.randn(B, H, W, c4, generator=g).to(DEV)
    w3 = (torch.randn(c4, 1, 1, k1, generator=g) / k1 ** 0.5 * torch.logspace(-1, 1, c4).view(-1, 1, 1, 1)).to(DEV)
    w1 = (torch.randn(mp, 1, 1, c4, generator=g) / c4 ** 0.5).to(DEV)
    sc3, sh3 = (torch.rand(c4, generator=g) + 0.5).to(DEV), torch.randn(c4, generator=g).to(DEV) * 0.2
    sc1, sh1 = (torch.rand(mp, generator=g) + 0.5).to(DEV), torch.randn(mp, generator=g).to(DEV) * 0.2
    s3 = ops.split_weight(w3.reshape(c4, k1), conv_shape=tuple(w3.shape), kind="f16x3")
    s1 = ops.split_weight(w1.reshape(mp, c4), conv_shape=tuple(w1.shape), kind="f16x3")
    X0 = ops.conv2d_nhwc(a, s3, scale=sc3, shift=sh3, R=R, relu=True)
    Y0 = ops.conv2d_nhwc(X0, s1, scale=sc1, shift=sh1, relu=True)
    blk = ops.BneckFused(s3, sc3, sh3, s1, sc1, sh1)
    X, Y1 = ops.bneck_fused(a, blk, R)
    dev = torch.device(DEV, torch.cuda.current_device())
    ops.check_range_flag(dev)
    # X: the same products in the same order; the epilogue's fma contraction may differ in the last bit (bneck2.hip: k-steps of 32 on
    # the 16x16x32 shape -- fp32-class, another order)
    assert float((X - X0).abs().max()) <= (1e-5 if blk.v2 else 2e-6) * float(X0.abs().max())
    # Y1: another summation order over c4 (accumulator order inside a 16-wide MFMA step)
    assert float((Y1 - Y0).abs().max()) <= 1e-5 * float(Y0.abs().max()) + 1e-6
    xr = torch.relu((a.double().cpu().view(-1, k1) @ w3.double().cpu().view(c4, k1).t()) * sc3.double().cpu() + sh3.double().cpu()
                    + R.double().cpu().view(-1, c4))
    yr = torch.relu((xr @ w1.double().cpu().view(mp, c4).t()) * sc1.double().cpu() + sh1.double().cpu())
    assert float((X.double().cpu().view(-1, c4) - xr).abs().max()) <= 3e-5 * float(xr.abs().max())
    assert float((Y1.double().cpu().view(-1, mp) - yr).abs().max()) <= 3e-5 * float(yr.abs().max())


def test_fused_pair_raises_the_range_flag():
    from gomatching_amd import ops
    dev = torch.device(DEV, torch.cuda.current_device())
    k1, c4, mp = 64, 256, 64
    s3 = ops.split_weight(torch.ones(c4, k1, device=DEV) / k1, conv_shape=(c4, 1, 1, k1), kind="f16x3")
    s1 = ops.split_weight(torch.ones(mp, c4, device=DEV) / c4, conv_shape=(mp, 1, 1, c4), kind="f16x3")
    one3, z3, one1, z1 = torch.ones(c4, device=DEV), torch.zeros(c4, device=DEV), torch.ones(mp, device=DEV), torch.zeros(mp, device=DEV)
    blk = ops.BneckFused(s3, one3, z3, s1, one1, z1)
    a, R = torch.ones(1, 9, 17, k1, device=DEV), torch.zeros(1, 9, 17, c4, device=DEV)
    ops.check_range_flag(dev)
    ops.bneck_fused(a, blk, R)
    ops.check_range_flag(dev)
    a2 = a.clone()
    a2[0, 3, 3, 5] = 7e4                                     # conv3's operand beyond fp16
    ops.bneck_fused(a2, blk, R)
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(dev)
    R2 = R.clone()
    R2[0, 4, 4, 100] = 1e5                                   # the block output (conv1's operand) beyond fp16
    ops.bneck_fused(a, blk, R2)
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(dev)
