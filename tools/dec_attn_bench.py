"""Interleaved A/B on one GPU: the decoder's self-attention blocks fused (csrc/dec_attn.hip) against the five-launch path
(projection GEMMs, attention core, out_proj + LayerNorm) at the bench's shape (8 frames x 100 queries x 25 points)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402

DEV = "cuda"
B, nq, P, E = 8, 100, 25, 256
Q = B * nq * P
g = torch.Generator().manual_seed(0)
in_w = (torch.randn(768, 256, generator=g) / 16).to(DEV)
in_b = (torch.randn(768, generator=g) * 0.1).to(DEV)
out_w = (torch.randn(256, 256, generator=g) / 16).to(DEV)
out_b = (torch.randn(256, generator=g) * 0.1).to(DEV)
gamma, beta = (torch.rand(256, generator=g) + 0.5).to(DEV), (torch.randn(256, generator=g) * 0.1).to(DEV)
x, pos = torch.randn(Q, 256, generator=g).to(DEV), torch.randn(Q, 256, generator=g).to(DEV)
wi = ops.split_weight(in_w, kind="f16x3")
wo = ops.split_weight(out_w, kind="f16x3")
qk, v_, qkv = ops.K256Linear(wi[:512], in_b[:512]), ops.K256Linear(wi[512:], in_b[512:]), ops.k256_linear(wi, in_b)
pl = ops.ProjLN(wo, out_b, gamma, beta)
intra, inter = ops.DecAttnBlock(wi, in_b, wo, out_b, gamma, beta, False, form=1), ops.DecAttnBlock(wi, in_b, wo, out_b, gamma, beta, True, form=1)
intra2, inter2 = ops.DecAttnBlock(wi, in_b, wo, out_b, gamma, beta, False, form=2), ops.DecAttnBlock(wi, in_b, wo, out_b, gamma, beta, True, form=2)
rw = ops.split_weight((torch.randn(384, 256, generator=g) / 16).to(DEV), kind="f16x3")
rb = (torch.randn(384, generator=g) * 0.1).to(DEV)
raw1 = ops.DecAttnBlock(wi, in_b, wo, out_b, gamma, beta, True, raw=(rw, rb), form=1)
raw2 = ops.DecAttnBlock(wi, in_b, wo, out_b, gamma, beta, True, raw=(rw, rb), form=2)
attn = torch.empty((Q, E), device=DEV)


def old_intra():
    a = ops.linear(x, qk, A2=pos)
    b = ops.linear(x, v_)
    f = a.view(-1)
    ops.mha_core(f, f[E:], b, attn, B * nq, 1, 8, 32, P, P, [P * 2 * E, 0, 2 * E, P * 2 * E, 0, 2 * E, P * E, 0, E, P * E, 0, E])
    return ops.proj_ln(attn, pl, x)


def old_inter():
    a = ops.linear(x, qkv)
    f = a.view(-1)
    ld = 3 * E
    ops.mha_core(f, f[E:], f[2 * E:], attn, B, P, 8, 32, nq, nq, [nq * P * ld, ld, P * ld] * 3 + [nq * P * E, E, P * E])
    return ops.proj_ln(attn, pl, x)


def new_intra():
    return ops.dec_attn(x, intra, B * nq, P, pos=pos)


def new_inter():
    return ops.dec_attn(x, inter, B * P, nq, inner=P)


def new_intra2():
    return ops.dec_attn(x, intra2, B * nq, P, pos=pos)


def new_inter2():
    return ops.dec_attn(x, inter2, B * P, nq, inner=P)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("max |fused - unfused| intra %.2e inter %.2e" % (float((new_intra() - old_intra()).abs().max()),
                                                      float((new_inter() - old_inter()).abs().max())))
for rnd in range(3):
    print("round %d: intra unfused %.1f us fused %.1f us | inter unfused %.1f us fused %.1f us" % (
        rnd, timeit(old_intra), timeit(new_intra), timeit(old_inter), timeit(new_inter)))
print("form 2 (16-token waves, two per SIMD): max |form 2 - form 1| intra %.2e inter %.2e" % (
    float((new_intra2() - new_intra()).abs().max()), float((new_inter2() - new_inter()).abs().max())))
for rnd in range(3):
    print("round %d: intra form 1 %.1f us form 2 %.1f us | inter %.1f / %.1f us | inter + raw %.1f / %.1f us" % (
        rnd, timeit(new_intra), timeit(new_intra2), timeit(new_inter), timeit(new_inter2),
        timeit(lambda: ops.dec_attn(x, raw1, B * P, nq, inner=P, raw_pos=pos)), timeit(lambda: ops.dec_attn(x, raw2, B * P, nq, inner=P, raw_pos=pos))))
fl = 2.0 * Q * 256 * 1024
print("fused intra: %.0f TFLOP/s of nn.Linear products; inter: %.0f" % (fl / timeit(new_intra) / 1e6, fl / timeit(new_inter) / 1e6))
