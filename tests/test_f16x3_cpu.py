"""The precision contract of the f16x3 contraction back-end (gomatching_amd/csrc/gemm_f16x3.hip), emulated in numpy:
x = x0 + x1 with x0 = fp16(x), x1 = fp16(x - x0); a.b ~= a0b0 + a0b1 + a1b0.  No GPU needed: this pins the arithmetic the
kernel implements (the GPU tests pin the kernel against fp64)."""
import numpy as np
import pytest


def split(x):
    x = np.asarray(x, np.float32)
    x0 = x.astype(np.float16)
    x1 = (x - x0.astype(np.float32)).astype(np.float16)
    return x0, x1


def scale_rows(w):
    """gom_split_f16x2: each row scaled by the power of two that puts its largest magnitude in [2^13, 2^14)."""
    mx = np.abs(w).max(axis=1)
    e = np.where(mx > 0, 14 - np.frexp(mx)[1], 0).astype(np.int32)
    return np.ldexp(w, e[:, None]).astype(np.float32), np.ldexp(np.float32(1), -e)


def gemm_f16x3(a, w):
    a0, a1 = split(a)
    ws, inv = scale_rows(w)
    w0, w1 = split(ws)
    f = lambda t: t.astype(np.float64)
    acc = f(a1) @ f(w0).T + f(a0) @ f(w1).T + f(a0) @ f(w0).T        # exact plane products, wide accumulation
    return acc * inv[None, :]


def test_two_planes_carry_22_bits_above_quarter():
    g = np.random.default_rng(0)
    x = (g.uniform(0.25, 60000.0, 200000) * g.choice([-1, 1], 200000)).astype(np.float32)
    x0, x1 = split(x)
    rel = np.abs(x0.astype(np.float64) + x1.astype(np.float64) - x.astype(np.float64)) / np.abs(x)
    assert rel.max() <= 2.0 ** -22


def test_absolute_floor_below_quarter():
    g = np.random.default_rng(1)
    x = g.uniform(-0.25, 0.25, 200000).astype(np.float32)
    x0, x1 = split(x)
    err = np.abs(x0.astype(np.float64) + x1.astype(np.float64) - x.astype(np.float64))
    assert err.max() <= 2.0 ** -25 + 1e-12                            # half a subnormal step of the second plane


def test_residual_is_exact_in_fp32():
    g = np.random.default_rng(2)
    x = (g.standard_normal(100000) * 37).astype(np.float32)
    x0 = x.astype(np.float16).astype(np.float32)
    r32 = x - x0                                                        # what the kernel computes (v_pk_add_f32)
    assert np.array_equal(r32.astype(np.float64), x.astype(np.float64) - x0.astype(np.float64))


@pytest.mark.parametrize("w_mag", [1e-9, 0.06, 1.0, 1e6])
def test_row_scaled_weights_keep_22_bits_at_any_magnitude(w_mag):
    g = np.random.default_rng(3)
    w = (g.standard_normal((64, 256)) * w_mag * np.logspace(-2, 2, 64)[:, None]).astype(np.float32)
    ws, inv = scale_rows(w)
    assert np.array_equal(ws.astype(np.float64) * inv[:, None], w.astype(np.float64))     # power-of-two scaling is exact
    w0, w1 = split(ws)
    assert np.isfinite(w0.astype(np.float32)).all()
    rec = (w0.astype(np.float64) + w1.astype(np.float64)) * inv[:, None]
    big = np.abs(ws) >= 4.0                                             # second plane normal: 22 significand bits
    assert (np.abs(rec - w)[big] / np.abs(w[big])).max() <= 2.0 ** -22
    assert np.abs(rec - w).max() <= 2.0 ** -25 * inv.max() + 2.0 ** -22 * np.abs(w).max()


@pytest.mark.parametrize("K", [64, 256, 1024])
def test_product_error_is_fp32_class(K):
    """Error of the three-term product against fp64, relative to sum |a||w|: <= 3 * 2^-22 by construction, and on
    N(0,1) data of the order of what re-ordering an fp32 accumulation changes."""
    g = np.random.default_rng(K)
    a = g.standard_normal((128, K)).astype(np.float32)
    w = (g.standard_normal((96, K)) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    got = gemm_f16x3(a, w)
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
    assert (np.abs(got - ref) / scale).max() <= 3 * 2.0 ** -22
    fp32 = (a @ w.T).astype(np.float64)                                 # numpy's fp32 GEMM, for scale
    assert np.abs(got - ref).max() <= 4 * max(np.abs(fp32 - ref).max(), 1e-7)


def test_out_of_range_activation_becomes_inf():
    x0, _ = split(np.array([70000.0], np.float32))
    assert np.isinf(x0.astype(np.float32)).all()                        # what the kernel's result check catches
