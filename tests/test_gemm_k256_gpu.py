"""Row-resident K = 256 GEMM (csrc/gemm_k256.hip) against the 128x128 tile kernel of the same f16x3 scheme: the plane products
run in the same order and the epilogue is the same fma, so the results must be IDENTICAL bits -- which is what lets the decoder
switch kernels by problem size without moving any golden.  Shapes: the decoder's Q-side layers
(/root/reference/third_party/adet/layers/deformable_transformer.py:386-422,470-488) + ragged / tiny M."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from gomatching_amd import ops
    ops.GEMM_MODE = "f16x3"
    return ops


@pytest.mark.parametrize("M,N", [(20000, 256), (20000, 384), (20000, 512), (20000, 768), (20000, 1024), (1, 32), (129, 64),
                                 (2500, 256), (4097, 96)])
@pytest.mark.parametrize("groups", [1, 2, 3])
def test_k256_equals_tile_kernel(M, N, groups):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn((M, 256), generator=g).to(DEV) * 1.7
    A2 = torch.randn((M, 256), generator=g).to(DEV)
    W = (torch.randn((N, 256), generator=g) * torch.logspace(-2, 1, N).view(-1, 1)).to(DEV)
    b = torch.randn((N,), generator=g).to(DEV)
    R = torch.randn((M, N), generator=g).to(DEV)
    sw = ops.split_weight(W, kind="f16x3")
    lin = ops.K256Linear(sw, b)
    if groups > N // 32:
        pytest.skip("more column groups than chunks")
    for kw in ({}, {"relu": True}, {"R": R}, {"A2": A2}, {"A2": A2, "R": R, "relu": True}):
        ref = ops.gemm(A, sw, bias=b, **kw)
        got = ops.linear(A, lin, groups=groups, **kw)
        assert torch.equal(got, ref), (kw.keys(), float((got - ref).abs().max()))
    if N >= 64:                                              # residual on the leading columns only (fused q|k|v style)
        rc = (N // 2) // 32 * 32
        ref = ops.gemm(A, sw, bias=b, R=R[:, :rc].contiguous() if False else R, r_cols=rc)
        got = ops.linear(A, lin, R=R, r_cols=rc, groups=groups)
        assert torch.equal(got, ref)
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


@pytest.mark.parametrize("lines", [0, 1])
@pytest.mark.parametrize("M,N,period", [(20000, 640, 2500), (4097, 96, 33), (129, 64, 32), (70000, 256, 0), (300, 512, 301),
                                        (8 * 9001, 640, 9001), (3 * 23333, 128, 23333)])   # long + periodic: frame-interleaved tiles
def test_k256_store_forms_and_periodic_residual(lines, M, N, period):
    """The kernel's two store forms (16-byte pieces with the row on the lane / whole 128-byte lines with the column on the
    lane, gom_gemm_k256_set_lines) return the tile kernel's bits -- with a residual on the leading columns, a PERIODIC one
    (row m adds R[m % period]: the encoder's position table, deepsolo.py encoder), ragged tails, column groups and `out` views."""
    ops = _ops()
    from gomatching_amd import lib
    g = torch.Generator().manual_seed(M + N + period)
    A = torch.randn((M, 256), generator=g).to(DEV) * 1.3
    W = (torch.randn((N, 256), generator=g) * torch.logspace(-2, 1, N).view(-1, 1)).to(DEV)
    b = torch.randn((N,), generator=g).to(DEV)
    sw = ops.split_weight(W, kind="f16x3")
    lin = ops.K256Linear(sw, b)
    rc = max(32, (N * 3 // 5) // 32 * 32)
    R = torch.randn((period or M, rc), generator=g).to(DEV)
    wide = torch.full((M, N + 64), 7.0, device=DEV)
    try:
        lib.load().gom_gemm_k256_set_lines(lines)
        lib.load().gom_gemm_k256_set_interleave(1)                    # the optional frame-interleaved tile order: same bits
        for groups in (1, 2):
            for kw in ({}, {"R": R, "r_cols": rc, "r_period": period}, {"R": R, "r_cols": rc, "r_period": period, "relu": True}):
                ref = ops.gemm(A, sw, bias=b, **kw)
                got = ops.linear(A, lin, groups=groups, **kw)
                assert torch.equal(got, ref), (groups, sorted(kw), float((got - ref).abs().max()))
        ops.linear(A, lin, groups=1, out=wide[:, 32:32 + N], R=R, r_cols=rc, r_period=period)
        assert torch.equal(wide[:, 32:32 + N], ops.gemm(A, sw, bias=b, R=R, r_cols=rc, r_period=period))
        assert float((wide[:, :32] - 7.0).abs().max()) == 0.0 and float((wide[:, 32 + N:] - 7.0).abs().max()) == 0.0
    finally:
        lib.load().gom_gemm_k256_set_lines(-1)
        lib.load().gom_gemm_k256_set_interleave(0)
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


def test_k256_views_slices_and_range_flag():
    """Row-strided operands (column slices of wider buffers), a weight ROW slice (the fused in_proj's q|k and v parts), no bias,
    an `out` view -- and the fp16 range contract: an activation beyond 65504 raises at the next check."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    wide = torch.randn((3000, 640), generator=g).to(DEV)
    A = wide[:, 128:384]
    W = torch.randn((768, 256), generator=g).to(DEV)
    sw = ops.split_weight(W, kind="f16x3")
    b = torch.randn((768,), generator=g).to(DEV)
    lin_qk, lin_v = ops.K256Linear(sw[:512], b[:512]), ops.K256Linear(sw[512:], None)
    out = torch.zeros((3000, 1024), device=DEV)
    ops.linear(A, lin_qk, out=out[:, 256:768], groups=1)
    assert torch.equal(out[:, 256:768], ops.gemm(A.contiguous(), sw[:512], bias=b[:512]))
    assert float(out[:, :256].abs().max()) == 0.0 and float(out[:, 768:].abs().max()) == 0.0
    assert torch.equal(ops.linear(A, lin_v, groups=1), ops.gemm(A.contiguous(), sw[512:]))
    dev = torch.device(DEV, torch.cuda.current_device())
    ops.check_range_flag(dev)
    big = torch.full((40, 256), 7e4, device=DEV)
    ops.linear(big, lin_v, groups=1)
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(dev)
    from gomatching_amd import lib
    L = lib.load()
    assert L.gom_gemm_k256_image_bytes(48, 256) == -1 and L.gom_gemm_k256_image_bytes(64, 512) == -1
    assert L.gom_gemm_k256_image_bytes(64, 256) == 2 * 33 * 1024


def test_kernel_choice_rule_is_a_speed_rule_only():
    """ops.linear picks the kernel from (M, N, second addend): whatever it picks, the bits are the tile kernel's."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    for M, N, a2 in ((5000, 256, False), (5000, 512, True), (5000, 512, False), (70000, 1024, False), (70000, 640, False), (70000, 256, False)):
        A = torch.randn((M, 256), generator=g).to(DEV)
        A2 = torch.randn((M, 256), generator=g).to(DEV) if a2 else None
        sw = ops.split_weight(torch.randn((N, 256), generator=g).to(DEV), kind="f16x3")
        b = torch.randn((N,), generator=g).to(DEV)
        assert torch.equal(ops.linear(A, ops.K256Linear(sw, b), A2=A2), ops.gemm(A, sw, bias=b, A2=A2))
    assert ops.k256_wins(20000, 256, False) and ops.k256_wins(20000, 768, True) and not ops.k256_wins(20000, 768, False)
    assert ops.k256_wins(297368, 1536, False) and ops.k256_wins(297368, 640, False) and not ops.k256_wins(297368, 256, False)
