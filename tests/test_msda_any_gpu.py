"""GPU: the general form of the native op (csrc/msda_any.hip) -- any heads / channels / levels / points, fp32 and fp64,
forward and backward -- against outputs and autograd gradients of the reference's own `ms_deform_attn_core_pytorch`
(tests/golden/msda_any.npz, oracle/gen_golden_msda_any.py), through `ops`, the `adet._C` stand-in and the dispatcher op."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = ["odd_f64", "odd_f32", "wide_f64", "one_f32", "ship_f64", "ship_f32"]
KEYS = ("_value", "_shapes", "_lsi", "_loc", "_w")


def _tol(x):
    return 1e-12 if x.dtype == torch.float64 else 2e-5


def _close(got, exp, what):
    exp = exp.to(got.device)
    assert got.dtype == exp.dtype and got.shape == exp.shape, what
    err = float((got - exp).abs().max())
    assert err <= _tol(exp) * max(1.0, float(exp.abs().max())), "%s: max|d| = %.3e" % (what, err)


def _case(name):
    g = golden("msda_any.npz")
    return g, [t(g[name + k]).to(DEV) for k in KEYS], t(g[name + "_gout"]).to(DEV)


@pytest.mark.parametrize("case", CASES)
def test_forward_and_backward_vs_the_reference_statement(case):
    from gomatching_amd import ops
    g, args, gout = _case(case)
    _close(ops.ms_deform_attn_forward(*args), t(g[case + "_out"]), case + " out")
    _close(ops.ms_deform_attn_forward_any(*args), t(g[case + "_out"]), case + " out (general kernel)")
    gv, gl, gw = ops.ms_deform_attn_backward(*args, gout)
    _close(gv, t(g[case + "_grad_value"]), case + " grad_value")
    _close(gl, t(g[case + "_grad_loc"]), case + " grad_sampling_loc")
    _close(gw, t(g[case + "_grad_w"]), case + " grad_attn_weight")
    # location / weight gradients carry no atomics: the same bits on every run
    gv2, gl2, gw2 = ops.ms_deform_attn_backward(*args, gout)
    assert torch.equal(gl, gl2) and torch.equal(gw, gw2)


def test_general_kernel_agrees_with_the_wave_per_query_kernel_on_the_shipped_shape():
    from gomatching_amd import ops
    g = golden("msda.npz")
    for case in ("enc", "dec", "oob"):
        args = [t(g[case + k]).to(DEV) for k in KEYS]
        fast, general = ops.ms_deform_attn_forward(*args), ops.ms_deform_attn_forward_any(*args)
        assert float((fast - general).abs().max()) <= 2e-6
        _close(general, t(g[case + "_out"]), case + " general kernel vs the reference fixture")


@pytest.mark.parametrize("case", ["odd_f64", "ship_f32"])
def test_reference_autograd_function_trains_through_the_stand_in(case):
    """The reference's MSDeformAttnFunction (third_party/adet/layers/ms_deform_attn.py:20-37), restated with `_C` bound to the
    stand-in: forward saves its inputs, backward returns (grad_value, None, None, grad_loc, grad_w, None)."""
    from gomatching_amd.compat import adet_C as _C

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, w, step):
            ctx.step = step
            ctx.save_for_backward(value, shapes, lsi, loc, w)
            return _C.ms_deform_attn_forward(value, shapes, lsi, loc, w, step)

        @staticmethod
        def backward(ctx, grad_output):
            gv, gl, gw = _C.ms_deform_attn_backward(*ctx.saved_tensors, grad_output.contiguous(), ctx.step)
            return gv, None, None, gl, gw, None

    g, args, gout = _case(case)
    for call in (lambda v, l, w: Fn.apply(v, args[1], args[2], l, w, 64),
                 lambda v, l, w: torch.ops.gomatching.ms_deform_attn_forward(v, args[1], args[2], l, w, 64)):
        v, l, w = (x.clone().requires_grad_(True) for x in (args[0], args[3], args[4]))
        out = call(v, l, w)
        out.backward(gout)
        _close(out.detach(), t(g[case + "_out"]), case + " out")
        _close(v.grad, t(g[case + "_grad_value"]), case + " grad_value")
        _close(l.grad, t(g[case + "_grad_loc"]), case + " grad_sampling_loc")
        _close(w.grad, t(g[case + "_grad_w"]), case + " grad_attn_weight")


def test_preconditions_and_dtype_refusal():
    from gomatching_amd import ops
    from gomatching_amd.compat import adet_C
    from gomatching_amd.lib import GomError, load
    g, args, gout = _case("odd_f32")
    with pytest.raises(RuntimeError, match="contiguous"):
        adet_C.ms_deform_attn_backward(*args, gout.transpose(0, 1), 64)
    with pytest.raises(RuntimeError, match="CUDA"):
        adet_C.ms_deform_attn_backward(*args, gout.cpu(), 64)
    with pytest.raises(GomError):
        ops.ms_deform_attn_backward(args[0], args[1], args[2], args[3], args[4].double(), gout)       # mixed dtypes
    with pytest.raises(GomError):
        ops.ms_deform_attn_forward(args[0], args[1], args[2], args[3][:, :, :2], args[4])             # heads disagree
    lib = load()
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    out = torch.empty(2, 11, 60, device=DEV)
    B, S, M, D = args[0].shape
    rc = lib.gom_ms_deform_attn_forward_any(7, p(args[0]), p(args[1]), p(args[2]), p(args[3]), p(args[4]), p(out), B, S, M, D,
                                            2, 11, 5, None)
    assert rc == 2                                                # GOM_ERR_UNSUPPORTED: neither float32 nor float64
    rc = lib.gom_ms_deform_attn_forward_any(0, p(args[0]), p(args[1]), p(args[2]), p(args[3]), p(args[4]), p(out), B, S, 0, D,
                                            2, 11, 5, None)
    assert rc == 1                                                # GOM_ERR_INVALID_ARG


def test_backward_at_encoder_size_sums_to_the_analytic_total():
    """Size-independent property at a full-size level pyramid (C2's 37 171 tokens): with every attention weight 1/(L*P) and a
    constant upstream gradient c, sum(grad_value) = c * (number of in-range corner weights) -- for locations strictly inside
    the map the four bilinear weights add to one, so the total is c * Lq * M * D exactly (up to summation order)."""
    from gomatching_amd import ops
    shapes = torch.tensor([[125, 223], [63, 112], [32, 56], [16, 28]], device=DEV)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, Lq, M, D, L, P = int(shapes.prod(1).sum()), 4096, 8, 32, 4, 4
    gen = torch.Generator(device=DEV).manual_seed(5)
    value = torch.randn(1, S, M, D, device=DEV, generator=gen)
    loc = torch.rand(1, Lq, M, L, P, 2, device=DEV, generator=gen) * 0.8 + 0.1
    w = torch.full((1, Lq, M, L, P), 1.0 / (L * P), device=DEV)
    gout = torch.full((1, Lq, M * D), 0.5, device=DEV)
    gv, gl, gw = ops.ms_deform_attn_backward(value, shapes, lsi, loc, w, gout)
    assert abs(float(gv.double().sum()) / (0.5 * Lq * M * D) - 1.0) < 1e-5
    # d/d(weight) of sample (l, p) = <grad_output, sampled value>: the forward with a one-hot weight says the same
    onehot = torch.zeros_like(w)
    onehot[..., 2, 1] = 1.0
    fwd = ops.ms_deform_attn_forward(value, shapes, lsi, loc, onehot).view(1, Lq, M, D)
    assert float((gw[..., 2, 1] - 0.5 * fwd.sum(-1)).abs().max()) < 1e-4
