"""3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS (csrc/conv3x3_patch.hip) against fp64 and against
the implicit-GEMM kernel it replaces for the ResNet bottlenecks' conv2 (Detectron2 BottleneckBlock; gom_lstmatcher.py:42-61)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, H, W, Cin, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    return x, w, sc, sh


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 19, 37, 64, 64), (1, 8, 16, 128, 128), (2, 33, 50, 256, 256), (1, 10, 20, 512, 512),
                                            (1, 7, 5, 64, 128), (3, 16, 32, 128, 64), (1, 250, 445, 64, 64)])
@pytest.mark.parametrize("relu", [True, False])
def test_patch_conv_vs_fp64_and_the_implicit_gemm_kernel(B, H, W, Cin, Cout, relu):
    from gomatching_amd import ops
    x, w, sc, sh = _case(B, H, W, Cin, Cout, seed=H * W + Cin)
    ref = F.conv2d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    sw = ops.split_weight(wd.reshape(Cout, -1), conv_shape=tuple(wd.shape), kind="f16x3")
    old = ops.CONV3_PATCH
    try:
        ops.CONV3_PATCH = True
        y = ops.conv2d_nhwc(xd, sw, scale=sc.to(DEV), shift=sh.to(DEV), relu=relu, stride=1, pad=1)
        ops.CONV3_PATCH = False
        z = ops.conv2d_nhwc(xd, sw, scale=sc.to(DEV), shift=sh.to(DEV), relu=relu, stride=1, pad=1)
    finally:
        ops.CONV3_PATCH = old
    torch.cuda.synchronize()
    ops.check_range_flag(DEV)
    err = float((y.permute(0, 3, 1, 2).cpu().double() - ref).abs().max())
    assert err <= 3e-5, err
    assert float((y - z).abs().max()) <= 8e-6                 # the same products; another k grouping
    if Cin == 64:
        pass                                                  # (single chunk: tap-major like the tile kernel, 32-wide k-steps)


def test_patch_conv_range_flag_and_untouched_neighbours():
    """An input beyond fp16's range raises the device flag (never a silent wrong result); a strided output view is not needed:
    the kernel writes exactly B*H*W*Cout floats."""
    from gomatching_amd import ops
    x, w, sc, sh = _case(1, 9, 17, 64, 64, seed=3)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    sw = ops.split_weight(wd.reshape(64, -1), conv_shape=tuple(wd.shape), kind="f16x3")
    old = ops.CONV3_PATCH
    ops.CONV3_PATCH = True
    try:
        ops.check_range_flag(DEV)
        xd[0, 4, 8, 5] = 7e4
        ops.conv2d_nhwc(xd, sw, relu=True, stride=1, pad=1)
        torch.cuda.synchronize()
        with pytest.raises(Exception):
            ops.check_range_flag(DEV)
        ops.check_range_flag(DEV)                             # cleared by the raise
    finally:
        ops.CONV3_PATCH = old
