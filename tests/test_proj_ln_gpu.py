"""out_proj + residual + LayerNorm in one launch (csrc/proj_ln.hip) against fp64 and against the two-launch path of the same
back-end (tile GEMM with the residual in its epilogue, then norm.hip's LayerNorm): deformable_transformer.py:258-264, 386-422."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(M, seed, wscale=0.06, xscale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((M, 256), generator=g) * xscale
    r = torch.randn((M, 256), generator=g)
    w = torch.randn((256, 256), generator=g) * wscale
    b = torch.randn((256,), generator=g) * 0.1
    ga = 1.0 + 0.2 * torch.randn((256,), generator=g)
    be = 0.1 * torch.randn((256,), generator=g)
    return x, r, w, b, ga, be


@pytest.mark.parametrize("M", [1, 31, 128, 129, 1000, 20000])
def test_proj_ln_vs_fp64_and_two_launches(M):
    from gomatching_amd import ops
    old, ops.GEMM_MODE = ops.GEMM_MODE, "f16x3"
    try:
        t = _case(M, seed=M)
        x, r, w, b, ga, be = [v.to(DEV) for v in t]
        sw = ops.split_weight(w, kind="f16x3")
        blk = ops.ProjLN(sw, b, ga, be)
        y = ops.proj_ln(x, blk, r)
        torch.cuda.synchronize()
        ops.check_range_flag(DEV)
        d = lambda v: v.double()
        ref = torch.nn.functional.layer_norm(d(t[0]) @ d(t[2]).T + d(t[3]) + d(t[1]), (256,), d(t[4]), d(t[5]), 1e-5)
        assert float((y.cpu().double() - ref).abs().max()) <= 2e-5
        z = ops.layernorm(ops.gemm(x, sw, bias=b, R=r), ga, be)
        assert float((y - z).abs().max()) <= 4e-6             # same pre-norm bits; the norm's sums run in another order
    finally:
        ops.GEMM_MODE = old


def test_proj_ln_strided_rows_wide_range_weights_in_place_and_range_flag():
    from gomatching_amd import ops
    old, ops.GEMM_MODE = ops.GEMM_MODE, "f16x3"
    try:
        M = 777
        x, r, w, b, ga, be = _case(M, seed=3)
        g = torch.Generator().manual_seed(4)
        w = w * torch.exp(torch.empty((256, 1)).uniform_(-6, 2, generator=g))     # row scales over 1e-3 .. 7
        d = lambda v: v.double()
        ref = torch.nn.functional.layer_norm(d(x) @ d(w).T + d(b) + d(r), (256,), d(ga), d(be), 1e-5)
        xb = torch.zeros((M, 640), device=DEV)
        xb[:, 128:384] = x.to(DEV)
        rb = torch.zeros((M, 512), device=DEV)
        rb[:, :256] = r.to(DEV)
        blk = ops.ProjLN(ops.split_weight(w.to(DEV), kind="f16x3"), None, ga.to(DEV), be.to(DEV))
        ref = torch.nn.functional.layer_norm(d(x) @ d(w).T + d(r), (256,), d(ga), d(be), 1e-5)
        out = rb[:, :256]
        ops.proj_ln(xb[:, 128:384], blk, rb[:, :256], out=out)                   # Y aliases R: rows are read before written
        torch.cuda.synchronize()
        assert float((out.cpu().double() - ref).abs().max()) <= 3e-5
        assert float(rb[:, 256:].abs().max()) == 0.0
        dev = torch.device(DEV, torch.cuda.current_device())
        ops.check_range_flag(dev)
        ops.proj_ln(torch.full((40, 256), 7e4, device=DEV), blk, torch.zeros((40, 256), device=DEV))
        with pytest.raises(Exception, match="fp16's range"):
            ops.check_range_flag(dev)
        from gomatching_amd import lib
        assert lib.load().gom_proj_ln_image_bytes(256, 512) == -1 and lib.load().gom_proj_ln_image_bytes(256, 256) == 8 * 32768
    finally:
        ops.GEMM_MODE = old


@pytest.mark.parametrize("M", [1, 130, 37171])
def test_no_residual_and_dot_form(M):
    """R = None: LayerNorm(x W^T + b) (enc_output + enc_output_norm, deformable_transformer.py:171-172); dot form: the class
    logit <LayerNorm(...), w> + b of every row without storing the rows (:175) -- against fp64, against the stored rows of the
    same kernel, and row-independent (a gathered subset gives the same bits)."""
    from gomatching_amd import ops
    old, ops.GEMM_MODE = ops.GEMM_MODE, "f16x3"
    try:
        t = _case(M, seed=M + 1)
        x, _, w, b, ga, be = [v.to(DEV) for v in t]
        blk = ops.ProjLN(ops.split_weight(w, kind="f16x3"), b, ga, be)
        g = torch.Generator().manual_seed(9)
        cw, cb = torch.randn((256,), generator=g).to(DEV) * 0.1, -1.25
        y = ops.proj_ln(x, blk, None)
        logit = ops.proj_ln_dot(x, blk, cw, cb)
        torch.cuda.synchronize()
        ops.check_range_flag(DEV)
        d = lambda v: v.double()
        ref = torch.nn.functional.layer_norm(d(t[0]) @ d(t[2]).T + d(t[3]), (256,), d(t[4]), d(t[5]), 1e-5)
        assert float((y.cpu().double() - ref).abs().max()) <= 2e-5
        assert float((logit.cpu().double() - (ref @ cw.cpu().double() + cb)).abs().max()) <= 2e-5
        assert float((logit - (y @ cw + cb)).abs().max()) <= 4e-6           # the same rows, the dot in another order
        assert torch.equal(y, ops.proj_ln(x, blk, torch.zeros_like(x)))       # + 0 residual: the same bits
        if M > 200:
            rows = torch.randperm(M, generator=g)[:100].to(DEV)
            assert torch.equal(ops.proj_ln(x[rows].contiguous(), blk, None), y[rows])
            assert torch.equal(ops.proj_ln_dot(x[rows].contiguous(), blk, cw, cb), logit[rows])
    finally:
        ops.GEMM_MODE = old


@pytest.mark.parametrize("M", [1, 63, 64, 65, 1000, 20000, 128 * 300 + 17])
def test_two_workgroups_per_cu_form_tile_edges_and_batch_invariance(M):
    """Launches run on 64-row tiles at two workgroups per CU (16x16x32 MFMA, k-steps of 32; the round-2 128-row form left the build
    in round 6): fp64 agreement around the tile edges, and a row's bits do not depend on the launch (batch invariance)."""
    from gomatching_amd import lib, ops
    g = torch.Generator().manual_seed(M)
    w = (torch.randn((256, 256), generator=g) * 0.06).to(DEV)
    b = (torch.randn((256,), generator=g) * 0.1).to(DEV)
    ga, be = (torch.rand((256,), generator=g) + 0.5).to(DEV), (torch.randn((256,), generator=g) * 0.1).to(DEV)
    blk = ops.ProjLN(ops.split_weight(w, kind="f16x3"), b, ga, be)
    x, r = torch.randn((M, 256), generator=g).to(DEV), torch.randn((M, 256), generator=g).to(DEV)
    new = ops.proj_ln(x, blk, r)
    part = ops.proj_ln(x[M // 2:].contiguous(), blk, r[M // 2:].contiguous())
    ops.check_range_flag(DEV)
    assert torch.equal(part, new[M // 2:])
    d = lambda t: t.double().cpu()
    ref = torch.nn.functional.layer_norm(d(x) @ d(w).T + d(b) + d(r), (256,), d(ga), d(be), 1e-5)
    assert float((d(new) - ref).abs().max()) <= 2e-5
