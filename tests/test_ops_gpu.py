"""GPU parity of every HIP kernel behind the C ABI against the CPU oracle / a plain fp32 torch-CPU
statement of the same op, on seeded inputs.  Tolerances are fp32-accumulation-order level."""
import math

import numpy as np
import ctypes

import pytest
import torch
import torch.nn.functional as F

from helpers import golden, t

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from gomatching_amd import ops
    return ops


def _close(a, b, atol, rtol=0.0, msg=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), "%s max|d|=%.3e (tol %.1e) at %s" % (
        msg, float(err.max()), atol, np.unravel_index(int(err.argmax()), tuple(err.shape)))


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(1, 1, 256), (25, 38, 256), (300, 384, 256), (513, 129, 1024), (100, 1024, 6400),
                                   (4097, 256, 256), (7, 8, 256), (130, 64, 64), (1000, 2, 256)])
def test_gemm_shapes(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    out = ops.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV))
    _close(out, F.linear(A, W, b), 2e-5, 1e-5, "gemm %s" % ((M, N, K),))


@pytest.mark.parametrize("M,N,K", [(1, 64, 256), (25, 38, 256), (300, 384, 256), (513, 129, 1024), (100, 1024, 6400),
                                   (4097, 256, 256), (130, 64, 64), (777, 1536, 256), (64, 100, 36)])
@pytest.mark.parametrize("kind", ["f16x3", "bf16x6"])
def test_gemm_split_shapes(M, N, K, kind):
    """Split-precision paths (two fp16 planes x 3 passes, three bf16 planes x 6 passes): same tolerance as the exact-fp32
    MFMA kernel (reference in float64)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g) * 3
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    sw = ops.split_weight(W.to(DEV), kind=kind)
    out = ops.gemm(A.to(DEV), sw, bias=b.to(DEV))
    ref = (A.double() @ W.double().t() + b.double()).float()
    _close(out, ref, 2e-5, 1e-5, "gemm %s %s" % (kind, (M, N, K)))
    if N > 40:                                       # row slices of the planes (in_proj q/k/v style)
        out2 = ops.gemm(A.to(DEV), sw[8:40], bias=b[8:40].to(DEV))
        _close(out2, ref[:, 8:40], 2e-5, 1e-5, "sliced planes")


@pytest.mark.parametrize("kind", ["f16x3", "bf16x6"])
def test_gemm_split_operands_on_rounding_ties(kind):
    """Operand values that sit EXACTLY on a rounding tie of the first plane (and whose residual is then exactly half a
    unit, incl. the smallest normal fp16): the split must still be exact -- a scalar-convert version of the split lost the
    low plane of such an element in the attention kernel (tools/flash_diag3.py).  Identity weights read the split back."""
    ops = _ops()
    import numpy as np
    K = 256
    g = np.random.default_rng(5)
    rows = []
    for e in range(-13, 6):                                       # binades 2^e .. 2^(e+1): fp16 ulp = 2^(e-10), bf16 2^(e-7)
        ulp = 2.0 ** (e - (10 if kind == "f16x3" else 7))
        m = g.integers(0, 1 << (10 if kind == "f16x3" else 7), size=K)
        x = (2.0 ** e + m * ulp + 0.5 * ulp) * g.choice([-1.0, 1.0], size=K)      # half-way between two plane-0 values
        rows.append(x.astype(np.float32))
    A = torch.from_numpy(np.stack(rows))
    assert torch.equal(A.double(), torch.from_numpy(np.stack(rows).astype(np.float64)))      # exactly representable
    eye = torch.eye(K)
    out = ops.gemm(A.to(DEV), ops.split_weight(eye.to(DEV), kind=kind)).cpu()
    rel = ((out - A).abs() / A.abs()).max()
    assert float(rel) <= 2.0 ** -21, float(rel)
    out_w = ops.gemm(eye.to(DEV), ops.split_weight(A[:, :K].contiguous().to(DEV), kind=kind)).cpu()     # ties as WEIGHTS
    rel_w = ((out_w.t() - A).abs() / A.abs()).max()
    assert float(rel_w) <= 2.0 ** -21, float(rel_w)


@pytest.mark.parametrize("kind", ["f16x3", "bf16x6"])
def test_gemm_split_epilogue_gather_and_extremes(kind):
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    M, N, K = 333, 256, 512
    A, A2 = torch.randn(M, K, generator=g), torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, sc = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    R = torch.randn(M, N, generator=g)
    sw = ops.split_weight(W.to(DEV), kind=kind)
    out = ops.gemm(A.to(DEV), sw, bias=b.to(DEV), scale=sc.to(DEV), A2=A2.to(DEV), R=R.to(DEV), relu=True)
    ref = F.relu(((A + A2).double() @ W.double().t()).float() * sc + b + R)
    _close(out, ref, 3e-5, 1e-5, "epilogue")
    rows = torch.randint(0, M, (77,), generator=g, dtype=torch.int64)
    out = ops.gemm(A.to(DEV), sw, bias=b.to(DEV), rows=rows.to(torch.int32).to(DEV))
    _close(out, (A[rows].double() @ W.double().t()).float() + b, 3e-5, 1e-5, "row gather")
    if kind == "bf16x6":
        # wide dynamic range: tiny and huge magnitudes in one dot product keep fp32-level RELATIVE accuracy (bf16 planes
        # carry fp32's exponent range; the fp16 planes do not, which is what the next block pins instead)
        A3 = torch.randn(64, K, generator=g) * torch.logspace(-6, 6, K)
        W3 = torch.randn(128, K, generator=g) * torch.logspace(3, -3, K)
        out = ops.gemm(A3.to(DEV), ops.split_weight(W3.to(DEV), kind=kind))
        ref = A3.double() @ W3.double().t()
        scale = (A3.double().abs() @ W3.double().abs().t())
        assert float(((out.cpu().double() - ref).abs() / scale).max()) < 2e-6
        return
    # f16x3 contract: weights of ANY magnitude (rows are power-of-two scaled before the split), activations up to 65504
    # with 2^-22 relative accuracy above ~0.25 and 3e-8 absolute accuracy below; beyond the range the flag is raised
    for w_mag, a_mag in ((1e-9, 1.0), (1e6, 1.0), (1.0, 1e4), (1.0, 1e-3)):
        A3 = torch.randn(64, K, generator=g) * a_mag
        W3 = torch.randn(128, K, generator=g) * w_mag * torch.logspace(-3, 3, 128).view(-1, 1)
        out = ops.gemm(A3.to(DEV), ops.split_weight(W3.to(DEV), kind=kind))
        ref = A3.double() @ W3.double().t()
        bound = 2e-6 * (A3.double().abs() @ W3.double().abs().t()) + 1e-7 * W3.double().abs().sum(1).view(1, -1)
        assert bool(((out.cpu().double() - ref).abs() <= bound).all()), (w_mag, a_mag)
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))       # nothing so far left the range
    ops.gemm(torch.full((8, K), 7e4).to(DEV), ops.split_weight(torch.ones(64, K).to(DEV), kind=kind))
    with pytest.raises(Exception, match="fp16's range"):
        ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


def test_f16x3_activation_split_equals_the_numpy_statement():
    """The in-kernel split of an activation (v_cvt_pk_f16_f32 + one v_fma_mix_f32 per residual, csrc/common.h) gives the planes
    of tests/test_f16x3_cpu.py's numpy statement: against an identity weight the product is x0 + x1 exactly (fp32 holds the
    22-bit sum), for ordinary values, fp16 rounding ties, values below fp16's normal range and the largest legal magnitudes."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    K = 64
    x = torch.randn(4096, K, generator=g) * torch.logspace(-9, 4.5, 4096).view(-1, 1)
    x[0, :] = torch.tensor([1.0 + 2.0 ** -11, 2048.0 + 1.0, 65504.0, -65504.0] * (K // 4))       # exact ties, the range's edge
    x[1, :] = torch.tensor([6e-8, -6e-8, 3e-5, 1e-30] * (K // 4))
    x = x.clamp(-65504.0, 65504.0)
    eye = torch.eye(K)
    out = ops.gemm(x.to(DEV), ops.split_weight(eye.to(DEV), kind="f16x3")).cpu()
    xn = x.numpy()
    x0 = xn.astype(np.float16)
    x1 = (xn - x0.astype(np.float32)).astype(np.float16)
    ref = x0.astype(np.float32) + x1.astype(np.float32)
    assert np.array_equal(out.numpy(), ref)
    lin = ops.K256Linear(ops.split_weight(torch.eye(256).to(DEV), kind="f16x3"), None)            # the row-resident kernels' split
    x4 = torch.cat([x, x, x, x], 1)
    out4 = ops.linear(x4.to(DEV), lin, groups=1).cpu()
    assert np.array_equal(out4.numpy(), np.concatenate([ref] * 4, 1))
    ops.check_range_flag(torch.device(DEV, torch.cuda.current_device()))


def test_f16x3_range_flag_is_raised_in_front_of_every_relu():
    """An activation beyond fp16's range splits into (Inf, -Inf) planes whose products sum to NaN, and fmaxf(NaN, 0) = 0: every
    f16x3 epilogue with a ReLU checks the PRE-activation value, so such a result is flagged, never silently zeroed -- the tile
    kernel's 16-byte and scalar epilogues, the split-K reduction, implicit-GEMM convolutions, the row-resident K = 256 kernel
    (both store forms)."""
    ops = _ops()
    from gomatching_amd import lib
    dev = torch.device(DEV, torch.cuda.current_device())
    K = 256

    def expect_flag(what):
        with pytest.raises(Exception, match="fp16's range"):
            ops.check_range_flag(dev)
        ops.check_range_flag(dev)                              # ... and the check re-arms the flag

    ops.check_range_flag(dev)
    big = torch.full((200, K), 7e4, device=DEV)
    for N in (64, 256, 66):                                    # N = 66: the scalar epilogue (N % 4 != 0)
        sw = ops.split_weight(torch.ones(N, K, device=DEV), kind="f16x3")
        out = ops.gemm(big, sw, relu=True)
        assert not bool(torch.isfinite(out).all()) or float(out.abs().max()) == 0.0   # what the hole looked like: all zeros
        expect_flag("gemm N=%d" % N)
        ops.gemm(torch.ones((200, K), device=DEV), sw, relu=True)
        ops.check_range_flag(dev)                              # in-range operands under the same epilogue: no flag
    # convolutions: 1x1 (the pointwise form of the tile kernel), 3x3, and the split-K 3x3 / 2 of input_proj[3]
    for Cin, Cout, k, stride, pad, H, W in ((64, 256, 1, 1, 0, 12, 20), (64, 64, 3, 1, 1, 12, 20), (2048, 256, 3, 2, 1, 8, 12)):
        x = torch.ones((2, H, W, Cin), device=DEV)
        w = torch.ones((Cout, k, k, Cin), device=DEV) / (Cin * k * k)
        sw = ops.split_weight(w.reshape(Cout, -1), conv_shape=tuple(w.shape), kind="f16x3")
        ops.conv2d_nhwc(x, sw, relu=True, stride=stride, pad=pad)
        ops.check_range_flag(dev)
        x[1, H // 2, W // 2, 3] = 7e4
        ops.conv2d_nhwc(x, sw, relu=True, stride=stride, pad=pad)
        expect_flag("conv %dx%d" % (k, k))
    # the row-resident kernel, 16-byte-store and whole-line-store forms
    W = torch.ones((64, K), device=DEV)
    lin = ops.K256Linear(ops.split_weight(W, kind="f16x3"), None)
    try:
        for lines in (0, 1):
            lib.load().gom_gemm_k256_set_lines(lines)
            ops.linear(big, lin, groups=1, relu=True)
            expect_flag("k256 lines=%d" % lines)
    finally:
        lib.load().gom_gemm_k256_set_lines(-1)


def test_gemm_epilogue_and_gather():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    M, N, K = 333, 256, 512
    A, A2 = torch.randn(M, K, generator=g), torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, sc = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    R = torch.randn(M, N, generator=g)
    out = ops.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), scale=sc.to(DEV), A2=A2.to(DEV), R=R.to(DEV), relu=True)
    ref = F.relu(F.linear(A + A2, W) * sc + b + R)
    _close(out, ref, 3e-5, 1e-5, "epilogue")
    rows = torch.randint(0, M, (77,), generator=g, dtype=torch.int64)
    out = ops.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), rows=rows.to(torch.int32).to(DEV))
    _close(out, F.linear(A[rows], W, b), 3e-5, 1e-5, "row gather")
    # strided views: column slices of a wider buffer for A, row slice of W, strided output
    big = torch.randn(M, 3 * K, generator=g).to(DEV)
    Wd = W.to(DEV)
    outbuf = torch.zeros(M, 2 * N, device=DEV)
    ops.gemm(big[:, K:2 * K], Wd[64:192], out=outbuf[:, N:N + 128])
    _close(outbuf[:, N:N + 128], F.linear(big[:, K:2 * K].cpu(), W[64:192]), 3e-5, 1e-5, "strided")
    assert float(outbuf[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(1, 1024, 1024), (26, 3072, 1024), (100, 1024, 6400), (128, 2048, 1024),
                                   (7, 600, 1024), (33, 1024, 256)])
def test_gemm_skinny_splitk(M, N, K):
    """Tracker / re-id head shapes take the split-K path inside ops.gemm; deterministic run to run."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Wd, bd, Rd = A.to(DEV), W.to(DEV), b.to(DEV), R.to(DEV)
    out = ops.gemm(Ad, Wd, bias=bd, R=Rd, relu=True)
    ref = F.relu((A.double() @ W.double().t()).float() + b + R)
    _close(out, ref, 2e-5, 1e-5, "splitk %s" % ((M, N, K),))
    assert torch.equal(out, ops.gemm(Ad, Wd, bias=bd, R=Rd, relu=True))
    rows = torch.randint(0, M, (max(M // 2, 1),), generator=g)
    out = ops.gemm(Ad, Wd, bias=bd, rows=rows.to(torch.int32).to(DEV))
    _close(out, (A[rows].double() @ W.double().t()).float() + b, 2e-5, 1e-5, "splitk gather")


@pytest.mark.parametrize("M,N,K", [(1, 1, 4), (5, 55, 1024), (59, 118, 1024), (26, 49, 1024), (128, 255, 260), (21, 3072, 1024), (7, 300, 36)])
def test_gemm_small_products(M, N, K):
    """One-wave-per-output kernel behind ops.gemm for the tracker's tiny products: fp64 reference, epilogue, row
    gather, strided operands, run-to-run identical bits."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 31 + N)
    A = torch.randn(M + 3, K + 8, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b, sc, R = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(M, N, generator=g)
    Ad, Wd = A.to(DEV), W.to(DEV)
    assert M * N <= ops.SMALL_GEMM_OUTPUTS
    out = ops.gemm(Ad[:M, :K], Wd, bias=b.to(DEV), scale=sc.to(DEV), R=R.to(DEV), relu=True, small=True)
    ref = F.relu((A[:M, :K].double() @ W.double().t()) * sc.double() + b.double() + R.double())
    _close(out, ref, 2e-5, 1e-5, "small gemm %s" % ((M, N, K),))
    assert torch.equal(out, ops.gemm(Ad[:M, :K], Wd, bias=b.to(DEV), scale=sc.to(DEV), R=R.to(DEV), relu=True, small=True))
    rows = torch.randint(0, M + 3, (M,), generator=g)
    out = ops.gemm(Ad[:, :K], Wd, rows=rows.to(torch.int32).to(DEV), small=True)
    _close(out, A[rows, :K].double() @ W.double().t(), 2e-5, 1e-5, "small gemm gather")
    out = ops.gemm(Ad[:M, :K], Ad[:M, :K], small=True)                  # association-logit form: X . X^T
    _close(out, A[:M, :K].double() @ A[:M, :K].double().t(), 3e-5 * math.sqrt(K), 1e-5, "x.xT")


def test_gemm_rejects_bad_args():
    ops = _ops()
    from gomatching_amd.lib import GomError
    A = torch.randn(8, 6, device=DEV)           # K not a multiple of 4
    W = torch.randn(4, 6, device=DEV)
    with pytest.raises(GomError):
        ops.gemm(A, W)


# ------------------------------------------------------------------------------------------ conv
def _conv_ref(x_nchw, w_oihw, stride, pad):
    return F.conv2d(x_nchw, w_oihw, None, stride=stride, padding=pad)


@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H,W", [
    (64, 64, 1, 1, 0, 13, 17), (256, 512, 1, 2, 0, 14, 18), (64, 64, 3, 1, 1, 13, 17), (128, 128, 3, 2, 1, 15, 19),
    (4, 64, 7, 2, 3, 37, 45), (512, 256, 3, 2, 1, 6, 9), (16, 32, 3, 1, 1, 9, 9)])
def test_conv_nhwc(Cin, Cout, k, stride, pad, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(Cin + Cout + k)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = _conv_ref(x, w, stride, pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    R = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + R)
    xd, wd = x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    Rd = R.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = ops.conv2d_nhwc(xd, wd, scale=sc.to(DEV), shift=sh.to(DEV), R=Rd, relu=True, stride=stride, pad=pad)
    _close(y.permute(0, 3, 1, 2), ref, 3e-5, 1e-5, "conv")
    for kind in ("f16x3", "bf16x6"):
        sw = ops.split_weight(wd.reshape(Cout, -1), conv_shape=tuple(wd.shape), kind=kind)
        y6 = ops.conv2d_nhwc(xd, sw, scale=sc.to(DEV), shift=sh.to(DEV), R=Rd, relu=True, stride=stride, pad=pad)
        _close(y6.permute(0, 3, 1, 2), ref, 3e-5, 1e-5, "conv " + kind)


def test_stem_preprocess_pool():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 30, 42, generator=g) * 255
    mean, std = [123.675, 116.280, 103.530], [58.395, 57.120, 57.375]
    x = ops.preprocess(img.to(DEV), mean, std)
    ref = (img - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    _close(x[..., :3].permute(0, 3, 1, 2), ref, 1e-6, 1e-6, "preprocess")
    assert float(x[..., 3].abs().max()) == 0.0
    f = torch.randn(2, 64, 15, 21, generator=g)
    y = ops.maxpool3x3s2(f.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), F.max_pool2d(f, 3, 2, 1))


# ------------------------------------------------------------------------------------------ norms
def test_layernorm_groupnorm():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    for D in (256, 1024):
        x, r = torch.randn(777, D, generator=g) * 3 + 1, torch.randn(777, D, generator=g)
        ga, be = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
        out = ops.layernorm(x.to(DEV), ga.to(DEV), be.to(DEV), residual=r.to(DEV))
        _close(out, F.layer_norm(x + r, (D,), ga, be, 1e-5), 2e-5, 1e-5, "layernorm %d" % D)
    B, H, W = 2, 9, 14
    x = torch.randn(B, 256, H, W, generator=g) * 2 + 0.5
    ga, be = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    ref = F.group_norm(x, 32, ga, be, 1e-5)
    S = H * W + 11
    buf = torch.zeros(B, S, 256, device=DEV)
    ops.groupnorm32_into(x.permute(0, 2, 3, 1).reshape(B, H * W, 256).contiguous().to(DEV), ga.to(DEV), be.to(DEV),
                         buf[0, 5:], S * 256)
    _close(buf[:, 5:5 + H * W].reshape(B, H, W, 256).permute(0, 3, 1, 2), ref, 2e-5, 1e-5, "groupnorm")
    assert float(buf[:, :5].abs().max()) == 0.0 and float(buf[:, 5 + H * W:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ MSDA
@pytest.mark.parametrize("case", ["enc", "dec", "oob"])
def test_msda_golden(case):
    """The reference's own output for the native op (fixture made by oracle/gen_golden.py)."""
    ops = _ops()
    g = golden("msda.npz")
    out = ops.ms_deform_attn_forward(t(g[case + "_value"]).to(DEV), t(g[case + "_shapes"]).to(DEV),
                                     t(g[case + "_lsi"]).to(DEV), t(g[case + "_loc"]).to(DEV),
                                     t(g[case + "_w"]).to(DEV), 64)
    _close(out, t(g[case + "_out"]), 2e-5, 0, "msda " + case)


def test_msda_larger_vs_oracle_and_prepare():
    from oracle import gom_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    shapes = [(23, 31), (12, 16), (6, 8), (3, 4)]
    S = sum(h * w for h, w in shapes)
    B, Lq = 2, 301
    ss = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    value = torch.randn(B, S, 8, 32, generator=g)
    raw = torch.randn(B * Lq, 384, generator=g)
    raw[:, :256] *= 3.0
    ref = torch.rand(B * Lq, 1, 2, generator=g) * 1.2 - 0.1
    loc, w = ops.msda_prepare(raw.to(DEV), ref.to(DEV), ss.to(DEV))
    off = raw[:, :256].view(B * Lq, 8, 4, 4, 2)
    norm = torch.stack([ss[:, 1], ss[:, 0]], -1)
    loc_ref = ref[:, None, :, None, :] + off / norm[None, None, :, None, :]
    w_ref = torch.softmax(raw[:, 256:].view(B * Lq, 8, 16), -1).view(B * Lq, 8, 4, 4)
    _close(loc, loc_ref, 1e-6, 1e-6, "prepare loc")
    _close(w, w_ref, 1e-6, 1e-5, "prepare w")
    out = ops.ms_deform_attn_forward(value.to(DEV), ss.to(DEV), lsi.to(DEV), loc.view(B, Lq, 8, 4, 4, 2),
                                     w.view(B, Lq, 8, 4, 4))
    exp = O.ms_deform_attn_forward(value, ss, lsi, loc_ref.view(B, Lq, 8, 4, 4, 2), w_ref.view(B, Lq, 8, 4, 4))
    _close(out, exp, 3e-5, 0, "msda vs oracle")
    # fused prepare+sample, value read in place from a wider buffer
    wide = torch.zeros(B * S, 640, device=DEV)
    wide[:, 384:] = value.view(B * S, 256).to(DEV)
    fused = ops.msda_fused(raw.to(DEV), ref.view(B * Lq, 2).to(DEV), wide[:, 384:], S * 640, ss.to(DEV),
                           lsi.to(DEV), B, Lq)
    _close(fused, exp.view(B * Lq, 256), 3e-5, 0, "fused msda vs oracle")


@pytest.mark.parametrize("shapes,B,scale", [([(23, 31), (12, 16), (6, 8), (3, 4)], 2, 2.0), ([(40, 72), (20, 36), (10, 18), (5, 9)], 3, 4.0),
                                            ([(17, 50), (9, 25), (5, 13), (3, 7)], 1, 9.0), ([(8, 16), (4, 8), (2, 4), (1, 2)], 2, 1.0),
                                            ([(125, 223), (63, 112), (32, 56), (16, 28)], 1, 2.5)])
def test_msda_encoder_window_kernel_is_bit_identical(shapes, B, scale):
    """The encoder form of the fused op (level-0 and level-1 queries served from LDS windows of the value map, csrc/msda.hip)
    against the lane-distributed kernel on the same inputs: the SAME BITS -- with offsets of a few pixels (everything inside the
    windows), with offsets far beyond the halo (the octet groups' global-memory path, counted), on maps that are not multiples of
    the 8 x 16 tiles and on the bench's pyramid (level-1 tiles gather their level-0 samples and window the rest)."""
    ops = _ops()
    from gomatching_amd import lib
    g = torch.Generator().manual_seed(int(scale * 10) + B)
    ss = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    S = int(ss.prod(1).sum())
    wide = torch.randn(B * S, 640, generator=g).to(DEV)
    raw = torch.randn(B * S, 384, generator=g)
    raw[:, :256] *= scale                                             # offsets in pixels of the sampled level
    raw = raw.to(DEV)
    ref = ops.encoder_reference_points(ss.to(DEV), lsi.to(DEV), S).repeat(B, 1).contiguous()
    L = lib.load()
    hw01 = tuple(shapes[0]) + tuple(shapes[1])
    plain = ops.msda_fused(raw, ref, wide[:, 384:], S * 640, ss.to(DEV), lsi.to(DEV), B, S)      # everything on the gather kernel
    counter = torch.zeros((1,), dtype=torch.int32, device=DEV)
    old = ops.MSDA_WINDOW_L1
    try:
        L.gom_msda_window_count_fallbacks(ctypes.c_void_p(counter.data_ptr()))
        for l1, hw in ((False, shapes[0]), (True, hw01)):             # level-0 windows alone | level-0 and level-1 windows
            ops.MSDA_WINDOW_L1 = l1
            counter.zero_()
            win = ops.msda_fused(raw, ref, wide[:, 384:], S * 640, ss.to(DEV), lsi.to(DEV), B, S, encoder_hw0=hw)
            assert torch.equal(win, plain), (l1, float((win - plain).abs().max()))
            groups = B * 8 * sum(-(-h // ty) * -(-w // tx) * (ty * tx // 8) for (h, w), (ty, tx) in zip(shapes[:2 if l1 else 1], ((8, 16), (8, 16))))
            slow = int(counter.item())
            assert 0 <= slow <= groups
            if scale <= 1.0:
                assert slow == 0, "offsets of ~1 pixel must stay inside the 5-pixel halo"
            if scale >= 9.0:
                assert slow > groups // 2, "offsets of ~9 pixels must leave the windows"
    finally:
        ops.MSDA_WINDOW_L1 = old
        L.gom_msda_window_count_fallbacks(None)


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("hd,Lq,Lk,outer,inner", [(32, 25, 25, 6, 1), (32, 12, 12, 2, 25), (32, 100, 100, 1, 3),
                                                  (128, 37, 90, 1, 1), (128, 5, 700, 1, 1), (32, 300, 300, 1, 2)])
def test_mha_core(hd, Lq, Lk, outer, inner):
    ops = _ops()
    g = torch.Generator().manual_seed(hd + Lq + Lk)
    heads = 8
    E = heads * hd
    nb = outer * inner
    q = torch.randn(nb, Lq, E, generator=g)
    k = torch.randn(nb, Lk, E, generator=g)
    v = torch.randn(nb, Lk, E, generator=g)
    qh = q.view(nb, Lq, heads, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    kh = k.view(nb, Lk, heads, hd).transpose(1, 2)
    vh = v.view(nb, Lk, heads, hd).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).transpose(1, 2).reshape(nb, Lq, E)
    out = torch.empty(nb, Lq, E, device=DEV)
    st = [inner * Lq * E, Lq * E, E, inner * Lk * E, Lk * E, E, inner * Lk * E, Lk * E, E, inner * Lq * E, Lq * E, E]
    ops.mha_core(q.to(DEV), k.to(DEV), v.to(DEV), out, outer, inner, heads, hd, Lq, Lk, st)
    _close(out, ref, 2e-5, 1e-5, "mha_core")


def test_mha_core_swapped_view():
    """The decoder's inter-instance attention: sequences over nq, batched over (b, point) without a transpose."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    B, nq, P, E, heads, hd = 2, 12, 25, 256, 8, 32
    qkv = torch.randn(B, nq, P, 3 * E, generator=g)
    x = qkv.permute(0, 2, 1, 3).reshape(B * P, nq, 3 * E)
    qh = x[..., :E].reshape(B * P, nq, heads, hd).transpose(1, 2) / math.sqrt(hd)
    kh = x[..., E:2 * E].reshape(B * P, nq, heads, hd).transpose(1, 2)
    vh = x[..., 2 * E:].reshape(B * P, nq, heads, hd).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).transpose(1, 2).reshape(B, P, nq, E).permute(0, 2, 1, 3)
    d = qkv.to(DEV)
    out = torch.empty(B, nq, P, E, device=DEV)
    ld = 3 * E
    st = [nq * P * ld, ld, P * ld] * 3 + [nq * P * E, E, P * E]
    flat = d.view(-1)
    ops.mha_core(flat, flat[E:], flat[2 * E:], out, B, P, heads, hd, nq, nq, st)
    _close(out, ref, 2e-5, 1e-5, "inter view")


# ------------------------------------------------------------------------------------------ glue
def test_positional_tables():
    from oracle import gom_oracle as O
    ops = _ops()
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / 128)
    H, W = 11, 17
    lvl = torch.randn(256)
    out = torch.empty(H * W, 256, device=DEV)
    ops.pos_encoding_into(dim_t.to(DEV), lvl.to(DEV), out, H, W)
    ref = O.pos_encoding_2d(torch.zeros(1, H, W, dtype=torch.bool))[0].flatten(1).t() + lvl
    _close(out, ref, 2e-6, 0, "pos2d")
    pts = torch.rand(333, 2)
    pe = ops.point_pos_embed(pts.to(DEV), dim_t.to(DEV))
    _close(pe, O.gen_point_pos_embed(pts, 256, 10000), 3e-6, 0, "point pos")
    delta = torch.randn(333, 4)
    ref2 = (delta[:, :2] + O.inverse_sigmoid(pts)).sigmoid()
    _close(ops.ref_sigmoid(delta.to(DEV), pts.to(DEV), 2), ref2, 1e-6, 0, "ref sigmoid 2")
    ref4 = (delta + O.inverse_sigmoid(pts).repeat(1, 2)).sigmoid()
    _close(ops.ref_sigmoid(delta.to(DEV), pts.to(DEV), 4), ref4, 1e-6, 0, "ref sigmoid 4")
    edge = torch.tensor([[0.0, 1.0], [1e-7, 1 - 1e-7], [0.5, 0.25]])
    _close(ops.ref_sigmoid(torch.zeros(3, 2, device=DEV), edge.to(DEV), 2), O.inverse_sigmoid(edge).sigmoid(), 1e-6, 0)


def test_topk_and_bezier():
    from oracle import gom_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    shapes = [(60, 90), (30, 45), (15, 23), (8, 12)]
    ss = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    S = int(ss.prod(1).sum())
    B, k = 3, 100
    logits = torch.randn(B * S, 1, generator=g)
    valid = ops.proposal_valid(ss.to(DEV), lsi.to(DEV), S)
    # oracle validity mask
    props = []
    for (H, W) in shapes:
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        props.append(torch.stack([(gx + 0.5) / W, (gy + 0.5) / H], -1).view(-1, 2))
    props = torch.cat(props, 0)
    vref = ((props > 0.01) & (props < 0.99)).all(-1)
    assert torch.equal(valid.cpu().bool(), vref)
    c0 = torch.tensor([0.37])
    idx = ops.topk_tokens(logits.to(DEV), B, S, k, valid=valid, invalid_logit=c0.to(DEV))
    masked = torch.where(vref[None], logits.view(B, S), c0)
    ref_idx = torch.topk(masked, k, dim=1)[1]
    got = idx.cpu().long()
    assert torch.equal(torch.gather(masked, 1, got), torch.gather(masked, 1, ref_idx))
    # without ties among the winners the index sets are identical
    for b in range(B):
        vals = torch.gather(masked, 1, ref_idx)[b]
        if len(torch.unique(vals)) == k:
            assert torch.equal(got[b], ref_idx[b])
    # small S (single chunk) and k == S
    l2 = torch.randn(2 * 37, 3, generator=g)
    i2 = ops.topk_tokens(l2.to(DEV), 2, 37, 37)
    assert torch.equal(i2.cpu().long(), torch.topk(l2[:, 0].view(2, 37), 37, dim=1)[1])
    # bezier reference points
    coord_raw = torch.randn(B, S, 8, generator=g)
    bern = O.bernstein_matrix(25)
    refs = ops.bezier_reference_points(coord_raw.to(DEV), idx, ss.to(DEV), lsi.to(DEV), bern.to(DEV), B, S, k, 25)
    lp = torch.log(props / (1 - props)).masked_fill(~vref[:, None], float("inf")).repeat(1, 4)
    unact = coord_raw + lp[None]
    sel = torch.gather(unact, 1, got.unsqueeze(-1).repeat(1, 1, 8)).sigmoid()
    exp = torch.matmul(bern, sel.view(B, k, 4, 2))
    _close(refs, exp, 2e-6, 0, "bezier refs")
    enc_ref = ops.encoder_reference_points(ss.to(DEV), lsi.to(DEV), S)
    _close(enc_ref, O.encoder_reference_points(shapes, torch.ones(1, 4, 2))[0, :, 0], 1e-7, 0, "enc ref")


# ------------------------------------------------------------------------------------------ detection
@pytest.mark.parametrize("with_re,nms_thr", [(True, 0.5), (False, 0.3)])
def test_detect_post_vs_oracle(with_re, nms_thr):
    from oracle import gom_oracle as O
    from helpers import mini_cfg
    ops = _ops()
    g = torch.Generator().manual_seed(41 + int(with_re))
    B, nq, P, V = 3, 60, 25, 38
    cfg = mini_cfg("icdar15", nq=nq)
    cfg.VIDEO_TEST.NMS_THRESH = nms_thr
    cfg.MODEL.ROI_HEADS.WITH_RESR = with_re
    thr = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST
    centers = torch.rand(B, nq, 1, 2, generator=g) * 0.8 + 0.1
    centers[:, ::3] = centers[:, 1::3][:, :centers[:, ::3].shape[1]]          # force overlapping boxes
    ctrl = (centers + (torch.rand(B, nq, P, 2, generator=g) - 0.5) * 0.1).clamp(0, 1)
    bd = torch.cat([ctrl - 0.02, ctrl + 0.02], -1).clamp(0, 1) + torch.rand(B, nq, P, 4, generator=g) * 0.01
    out = {
        "pred_logits": torch.randn(B, nq, P, 1, generator=g) * 2 - 0.5,
        "pred_text_logits": torch.randn(B, nq, P, V, generator=g),
        "pred_ctrl_points": ctrl, "pred_bd_points": bd,
        "query_features": torch.randn(B, nq, P, 256, generator=g),
    }
    out["pred_logits"][0] = -20.0                                  # an empty frame
    re = torch.randn(B, nq, P, 1, generator=g) * 2 - 1.0 if with_re else None
    if re is not None:
        re[0] = -20.0
    hw = (96, 128)
    det = O.detection(cfg, out, re, [hw] * B)
    props = O.proposals_with_nms(cfg, det)
    recs = ops.argmax_rows(out["pred_text_logits"].reshape(-1, V).to(DEV))
    assert torch.equal(recs.cpu().long(), out["pred_text_logits"].reshape(-1, V).argmax(-1))
    r = ops.detect_post(out["pred_logits"].reshape(-1, 1).to(DEV),
                        re.reshape(-1, 1).to(DEV) if re is not None else None,
                        ctrl.to(DEV), bd.to(DEV), recs, B, nq, P, hw[0], hw[1], thr, nms_thr, thr)
    cnt = r["count"].cpu().tolist()
    assert cnt[0] == 0
    for b in range(B):
        p = props[b].select(props[b]["objectness_logits"] > thr)
        n = len(p)
        assert cnt[b] == n, (b, cnt[b], n)
        if n == 0:
            continue
        _close(r["scores"][b, :n], p["scores"], 1e-6, 0, "scores")
        _close(r["boxes"][b, :n], p["proposal_boxes"], 1e-4, 0, "boxes")
        _close(r["ctrl"][b, :n], p["ctrl_points"], 1e-4, 0, "ctrl")
        _close(r["bd"][b, :n], p["bd"], 1e-4, 0, "bd")
        assert torch.equal(r["recs"][b, :n].cpu(), p["recs"])
        # keep_idx points at the right query rows
        qf = out["query_features"].reshape(B * nq, P, 256)[r["keep_idx"][b, :n].cpu().long()]
        assert torch.equal(qf, p["query_features"])


# ------------------------------------------------------------------------------------------ tracker pieces
def test_asso_activate_and_track_score():
    ops = _ops()
    g = torch.Generator().manual_seed(51)
    n_t = [4, 0, 7, 5]
    N, n_k = sum(n_t), 5
    logits = torch.randn(n_k, N, generator=g) * 3
    offs = torch.tensor([0, 4, 4, 11, 16], dtype=torch.int32)
    act = ops.asso_activate(logits.to(DEV), offs.to(DEV), 4)
    ref = torch.cat([torch.cat([a, a.new_zeros((n_k, 1))], 1).softmax(1)[:, :-1] for a in logits.split(n_t, 1)], 1)
    _close(act, ref, 1e-6, 1e-5, "activate")
    # track score: frames 0..2 are history (k = 3)
    boxes = torch.rand(N, 2, generator=g) * 80
    boxes = torch.cat([boxes, boxes + torch.rand(N, 2, generator=g) * 30 + 5], 1)
    ids = torch.tensor([3, 5, 9, 2, 5, 3, 9, 11, 2, 7, 14])
    nonk = list(range(0, 11))
    k_inds = list(range(11, 16))
    uniq = torch.unique(ids)
    M, Np = len(uniq), len(ids)
    col_of = torch.searchsorted(uniq, ids)
    id_inds = (uniq[None, :] == ids[:, None]).float()
    last = (id_inds * torch.arange(Np)[:, None]).max(dim=0)[1]
    dts = torch.tensor([2.0] * 4 + [0.0] * 7)
    decay = 0.9 ** dts
    meta = torch.tensor(nonk + col_of.tolist() + last.tolist() + k_inds, dtype=torch.int32)
    img_w, img_h = 128.0, 96.0
    traj = ops.track_score(act, meta.to(DEV), decay.to(DEV), boxes.to(DEV), img_w, img_h, n_k, Np, M, True, 1.0)
    from oracle import gom_oracle as O
    nb = boxes.clone()
    nb[:, [0, 2]] /= img_w
    nb[:, [1, 3]] /= img_h
    kb, ob = nb[k_inds], nb[nonk]
    tr = torch.mm(ref[:, nonk] * decay[None], id_inds)
    tr = torch.max(tr, O.pairwise_iou(kb, ob[last]))
    k_ct = (kb[:, :2] + kb[:, 2:]) / 2
    k_s = ((kb[:, 2:] - kb[:, :2]) ** 2).sum(1)
    n_ct = (ob[:, :2] + ob[:, 2:]) / 2
    dist = ((k_ct[:, None] - n_ct[None]) ** 2).sum(2) / (k_s[:, None] + 1e-8)
    va = torch.mm((dist < 1.0).float(), id_inds).clamp_(max=1.0).bool()
    tr[~va] = 0
    _close(traj, tr, 2e-6, 1e-5, "track score")
    rows = torch.tensor([5, 0, 3, 3], dtype=torch.int32)
    src = torch.randn(9, 1024, generator=g)
    assert torch.equal(ops.gather_rows(src.to(DEV), rows.to(DEV)).cpu(), src[rows.long()])


def test_gemm_periodic_residual_equals_broadcast_table():
    """gom_gemm_f32_f16x3_rp: residual row m % period (the encoder's [S, 384] position table shared by the frames of a step)
    gives the bits of the same GEMM fed with the table broadcast to every frame; columns beyond r_cols untouched."""
    ops = _ops()
    old, ops.GEMM_MODE = ops.GEMM_MODE, "f16x3"
    try:
        g = torch.Generator().manual_seed(21)
        S, B, N = 1237, 3, 640
        A = torch.randn((B * S, 256), generator=g).to(DEV)
        W = ops.split_weight(torch.randn((N, 256), generator=g).to(DEV) * 0.1, kind="f16x3")
        b = torch.randn((N,), generator=g).to(DEV)
        table = torch.randn((S, 384), generator=g).to(DEV)
        ref = ops.gemm(A, W, bias=b, R=table.repeat(B, 1), r_cols=384)
        got = ops.gemm(A, W, bias=b, R=table, r_cols=384, r_period=S)
        assert torch.equal(got, ref)
        with pytest.raises(AssertionError):
            ops.gemm(A, torch.randn((N, 256), device=DEV), R=table, r_period=S)
    finally:
        ops.GEMM_MODE = old


def test_ref_update_equals_the_three_launches_it_replaces():
    """N = 2 linear + ref_sigmoid + point_pos_embed (deformable_transformer.py:484-488, :470-473) as one launch: the same
    reference points up to the dot product's summation order, the same embedding arithmetic on them."""
    from gomatching_amd import ops
    torch.manual_seed(3)
    Q = 2500 * 3 + 7
    h = torch.randn(Q, 256, device=DEV)
    W3, b3 = torch.randn(2, 256, device=DEV) * 0.05, torch.randn(2, device=DEV) * 0.1
    ref = torch.rand(Q, 2, device=DEV)
    ref[:5] = torch.tensor([[0.0, 1.0], [1.0, 0.0], [1e-7, 0.5], [0.5, 1 - 1e-7], [0.3, 0.3]], device=DEV)   # the eps clamps
    dim_t = 10000 ** (2 * torch.div(torch.arange(128, dtype=torch.float32), 2, rounding_mode="trunc") / 128)
    dim_t = dim_t.to(DEV)
    exp_ref = ops.ref_sigmoid(ops.gemm(h, W3, bias=b3), ref, 2)
    for scale in (None, (0.75, 0.9)):
        new_ref, pos = ops.ref_update(h, (W3, b3), ref, dim_t, scale)
        assert float((new_ref - exp_ref).abs().max()) <= 2e-6
        scaled = new_ref if scale is None else ops.scale_xy_(new_ref.clone(), *scale)
        assert torch.equal(pos, ops.point_pos_embed(scaled, dim_t))          # same arithmetic on the same points: same bits
    only_ref, none = ops.ref_update(h, (W3, b3), ref, dim_t, want_pos=False)
    assert none is None and torch.equal(only_ref, new_ref)
    wide = torch.randn(Q, 512, device=DEV)                                   # a column slice of a wider buffer (row stride 512)
    got, _ = ops.ref_update(wide[:, 256:], (W3, b3), ref, dim_t, want_pos=False)
    exp, _ = ops.ref_update(wide[:, 256:].contiguous(), (W3, b3), ref, dim_t, want_pos=False)
    assert torch.equal(got, exp)
