"""Golden fixtures for the GENERAL form of the native op: any heads / channels / levels / points, fp32 and fp64, forward AND
backward (container-only).

TEST INFRASTRUCTURE.  Run in the build container, where /root/reference exists:

    python -m oracle.gen_golden_msda_any        # writes tests/golden/msda_any.npz

The reference's compiled op dispatches on the value dtype (third_party/adet/layers/csrc/DeformAttn/ms_deform_attn_cuda.cu:
64 forward, :134 backward: AT_DISPATCH_FLOATING_TYPES) and takes every shape from its tensors (:41-48); its CUDA sources do
not build here.  What IS runnable is the reference's own pure-PyTorch statement of the same op,
`ms_deform_attn_core_pytorch` (third_party/adet/layers/ms_deform_attn.py:40-60), which the reference keeps "for debug and
test" of exactly that kernel: its outputs, and its autograd gradients with respect to value / sampling locations / attention
weights for a random upstream gradient, are stored here.  The oracle's restatement (oracle/gom_oracle.py
ms_deform_attn_forward, differentiated by autograd) is checked against them as they are made.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import gom_oracle as O                               # noqa: E402
from oracle import ref_shim                                      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name: (B, heads, channels, level shapes, queries, points, dtype, location range)
CASES = {
    "odd_f64": (2, 3, 20, [(5, 7), (3, 4)], 11, 5, torch.float64, (-0.3, 1.3)),
    "odd_f32": (2, 3, 20, [(5, 7), (3, 4)], 11, 5, torch.float32, (-0.3, 1.3)),
    "wide_f64": (1, 2, 80, [(4, 6), (2, 3), (1, 2)], 9, 2, torch.float64, (0.0, 1.0)),
    "one_f32": (3, 1, 7, [(6, 5)], 13, 1, torch.float32, (-0.2, 1.2)),
    "ship_f64": (1, 8, 32, [(6, 9), (3, 5), (2, 3), (1, 2)], 17, 4, torch.float64, (-0.1, 1.1)),
    "ship_f32": (1, 8, 32, [(6, 9), (3, 5), (2, 3), (1, 2)], 17, 4, torch.float32, (-0.1, 1.1)),
}


def make_inputs(name):
    B, M, D, shapes, Lq, P, dtype, (lo, hi) = CASES[name]
    g = torch.Generator().manual_seed(sum(ord(c) for c in name))
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, M, D, generator=g, dtype=torch.float64).to(dtype)
    loc = (torch.rand(B, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * (hi - lo) + lo).to(dtype)
    w = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g, dtype=torch.float64), -1).view(B, Lq, M, L, P).to(dtype)
    gout = torch.randn(B, Lq, M * D, generator=g, dtype=torch.float64).to(dtype)
    ss = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    return value, ss, lsi, loc, w, gout, shapes


def grads_of(fn, value, loc, w, gout):
    value, loc, w = (t.detach().clone().requires_grad_(True) for t in (value, loc, w))
    out = fn(value, loc, w)
    gv, gl, gw = torch.autograd.grad(out, (value, loc, w), gout)
    return out.detach(), gv, gl, gw


def main():
    msda = ref_shim.load("adet.layers.ms_deform_attn")
    store = {}
    for name in CASES:
        value, ss, lsi, loc, w, gout, shapes = make_inputs(name)
        ref = grads_of(lambda v, l_, w_: msda.ms_deform_attn_core_pytorch(v, shapes, l_, w_), value, loc, w, gout)
        mine = grads_of(lambda v, l_, w_: O.ms_deform_attn_forward(v, ss, lsi, l_, w_), value, loc, w, gout)
        tol = 1e-12 if value.dtype == torch.float64 else 2e-5
        for what, a, b in zip(("out", "grad_value", "grad_loc", "grad_w"), ref, mine):
            d = float((a - b).abs().max())
            print("msda_any/%s %-10s oracle-vs-reference max|d| = %.3e" % (name, what, d))
            assert d <= tol * max(1.0, float(a.abs().max())), (name, what, d)
        store.update({name + "_value": value.numpy(), name + "_shapes": ss.numpy(), name + "_lsi": lsi.numpy(),
                      name + "_loc": loc.numpy(), name + "_w": w.numpy(), name + "_gout": gout.numpy(),
                      name + "_out": ref[0].numpy(), name + "_grad_value": ref[1].numpy(),
                      name + "_grad_loc": ref[2].numpy(), name + "_grad_w": ref[3].numpy()})
    np.savez_compressed(os.path.join(GOLD, "msda_any.npz"), **store)
    print("wrote", os.path.join(GOLD, "msda_any.npz"))


if __name__ == "__main__":
    main()
