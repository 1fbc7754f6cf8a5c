"""Per-shape timing of every GEMM-class launch on the BASELINE workload's transformer (encoder shapes, the decoder's Q-side
GEMMs north_star names, the fused FFN block), f16x3 back-end.  Stand-alone: HIP-event medians.  Under rocprofv3 the same run
gives the per-shape kernel durations that tools/gemm_shapes_csv.py reduces to profiles/r02_gemm_shapes.csv:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_shapes -- python3 tools/gemm_shapes.py
    python tools/gemm_shapes_csv.py gpurun_out/prof_shapes/*/*kernel_trace.csv gpurun_out/gemm_shapes_order.json"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
ops.GEMM_MODE = "f16x3"
S8, Q = 37171 * 8, 8 * 100 * 25
REP = int(os.environ.get("REP", 12))
# (label, M, N, K, residual columns, A2)
SHAPES = [("enc offsets|logits|value N=640 (+pos table on 384 cols)", S8, 640, 256, 384, False),
          ("enc out_proj N=256 (+residual)", S8, 256, 256, 256, False),
          ("enc/dec value_proj x6 N=1536", S8, 1536, 256, 0, False),
          ("enc_output N=256", S8, 256, 256, 0, False),
          ("dec Q-side N=256 K=256 (out_proj / qpos / v / MLP, +residual)", Q, 256, 256, 256, False),
          ("dec Q-side N=384 K=256 (cross offsets|logits, A2 = qpos)", Q, 384, 256, 0, True),
          ("dec Q-side N=512 K=256 (intra q|k, A2 = qpos)", Q, 512, 256, 0, True),
          ("dec Q-side N=768 K=256 (inter q|k|v)", Q, 768, 256, 0, False),
          ("dec Q-side N=1024 K=256 (linear1, three-launch path)", Q, 1024, 256, 0, False),
          ("dec Q-side N=256 K=1024 (linear2, three-launch path)", Q, 256, 1024, 256, False)]
g = torch.Generator().manual_seed(0)
order = []


def run(label, fn, flops, kernel):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REP):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    order.append({"label": label, "kernel": kernel, "launches": REP + 2, "flops": flops})
    print("%-62s %9.1f us  %6.1f TFLOP/s  (%.3f of 833)" % (label, us, flops / us / 1e6, flops / us / 1e6 / 833.3), flush=True)


for label, M, N, K, rc, a2 in SHAPES:
    A = torch.randn((M, K), generator=g).to(dev)
    W = ops.split_weight((torch.randn((N, K), generator=g) / K ** 0.5).to(dev), kind="f16x3")
    b = torch.randn((N,), generator=g).to(dev)
    # the encoder's position table is read periodically (one [S, 384] table shared by the 8 frames: DeepSolo.geometry)
    period = M // 8 if (rc and rc < N) else 0
    R = torch.randn((period or M, rc), generator=g).to(dev) if rc else None
    A2 = torch.randn((M, K), generator=g).to(dev) if a2 else None
    out = torch.empty((M, N), device=dev)
    # the product's own dispatch: ops.linear picks the row-resident K = 256 kernel where it measured faster (ops.k256_wins)
    lin = ops.k256_linear(W, b) if K == 256 else (W, b)
    kern = "gemm_k256_kernel" if isinstance(lin, ops.K256Linear) and ops.k256_wins(M, N, a2) else "gemm_f16x3_kernel"
    run(label, lambda: ops.linear(A, lin, R=R, r_cols=rc if rc else None, r_period=period, A2=A2, out=out), 2.0 * M * N * K, kern)
    del A, W, R, A2, out
for label, M in (("enc out_proj + residual + LayerNorm fused (proj_ln)", S8), ("dec out_proj + residual + LayerNorm fused (stand-alone; in the product inside the tail launch)", Q)):
    w = (torch.randn((256, 256), generator=g) * 0.06).to(dev)
    blk = ops.ProjLN(ops.split_weight(w, kind="f16x3"), torch.randn((256,), generator=g).to(dev),
                     torch.ones((256,), device=dev), torch.zeros((256,), device=dev))
    x = torch.randn((M, 256), generator=g).to(dev)
    r = torch.randn((M, 256), generator=g).to(dev)
    y = torch.empty_like(x)
    run(label, lambda: ops.proj_ln(x, blk, r, out=y), 2.0 * M * 256 * 256, "proj_ln2_kernel")
    del x, r, y
for label, M in (("enc FFN block fused (linear1+ReLU+linear2+residual+LayerNorm)", S8), ("dec FFN block fused (stand-alone; in the product inside the tail launch)", Q)):
    F = 1024
    w1 = (torch.randn((F, 256), generator=g) * 0.05).to(dev); b1 = torch.randn((F,), generator=g).to(dev) * 0.1
    w2 = (torch.randn((256, F), generator=g) * 0.05).to(dev); b2 = torch.randn((256,), generator=g).to(dev) * 0.1
    ffn = ops.FusedFFN(w1, b1, w2, b2, torch.ones((256,), device=dev), torch.zeros((256,), device=dev))
    x = torch.randn((M, 256), generator=g).to(dev)
    y = torch.empty_like(x)
    run(label, lambda: ops.ffn_fused_ln(x, ffn, out=y), 4.0 * M * 256 * F, "ffn_fused_kernel")
# ---- round 5: the decoder layer's launches as the product issues them, and the res4 bottleneck pair ----
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from test_dec_tail_gpu import _case as _tail_case                # noqa: E402
x, ffn_w, coord, qpos_w, ref, dim_t = _tail_case(Q, 1024, seed=3)
dv = lambda t: t.to(dev)
wo, bo = dv(torch.randn((256, 256), generator=g) / 16), dv(torch.randn((256,), generator=g) * 0.1)
tail = ops.DecTail(tuple(dv(v) for v in ffn_w), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos_w], dv(dim_t),
                   proj_w=(wo, bo, torch.ones((256,), device=dev), torch.zeros((256,), device=dev)))
samp, X, R = dv(torch.randn((Q, 256), generator=g)), dv(x), dv(ref)
run("dec layer tail: cross out_proj + norm, FFN + norm3, ctrl_point_coord + refinement, next ref_point_head (one launch)",
    lambda: ops.dec_tail(samp, tail, R, want_qpos=True, residual=X), 2.0 * Q * 256 * (256 + 2 * 1024 + 4 * 256 + 2),
    "dec_tail2_kernel" if tail.form == 2 else "dec_tail_kernel")
in_w, in_b = dv(torch.randn((768, 256), generator=g) / 16), dv(torch.randn((768,), generator=g) * 0.1)
ones, zeros = torch.ones((256,), device=dev), torch.zeros((256,), device=dev)
rw, rb = dv(torch.randn((384, 256), generator=g) / 16), dv(torch.randn((384,), generator=g) * 0.1)
intra = ops.DecAttnBlock(in_w, in_b, wo, bo, ones, zeros, False)
inter = ops.DecAttnBlock(in_w, in_b, wo, bo, ones, zeros, True, raw=(rw, rb))
pos = dv(torch.randn((Q, 256), generator=g))
run("dec intra-instance block (in_proj, 8 x 32 attention over 25 points, out_proj, norm: one launch)",
    lambda: ops.dec_attn(X, intra, 800, 25, pos=pos), 2.0 * Q * 256 * 1024, "dec_attn2_kernel" if intra.form == 2 else "dec_attn_kernel")
run("dec inter-instance block + cross offsets|logits (one launch)",
    lambda: ops.dec_attn(X, inter, 200, 100, inner=25, raw_pos=pos), 2.0 * Q * 256 * (1024 + 384), "dec_attn2_kernel" if inter.form == 2 else "dec_attn_kernel")
del samp, X, R, pos
k1, c4, mp, hw = 256, 1024, 256, (63, 112)
a = dv(torch.randn(8, hw[0], hw[1], k1, generator=g).abs())
Rr = dv(torch.randn(8, hw[0], hw[1], c4, generator=g))
s3 = ops.split_weight(dv(torch.randn(c4, k1, generator=g) / k1 ** 0.5), conv_shape=(c4, 1, 1, k1), kind="f16x3")
s1 = ops.split_weight(dv(torch.randn(mp, c4, generator=g) / c4 ** 0.5), conv_shape=(mp, 1, 1, c4), kind="f16x3")
blk2 = ops.BneckFused(s3, torch.ones(c4, device=dev), torch.zeros(c4, device=dev), s1, torch.ones(mp, device=dev), torch.zeros(mp, device=dev))
run("res4 bottleneck pair: conv3 256->1024 + BN + residual + ReLU, next conv1 1024->256 + BN + ReLU (one launch, M = 56 448)",
    lambda: ops.bneck_fused(a, blk2, Rr), 2.0 * 8 * hw[0] * hw[1] * c4 * (k1 + mp), "bneck2_kernel")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(order, open("gpurun_out/gemm_shapes_order.json", "w"), indent=1)
