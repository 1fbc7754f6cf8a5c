"""A/B in ONE process: gom_gemm_f32_bf16x6 (fp32 A, split in the loop) vs gom_gemm_planes_bf16x6 (pre-split A, LDS-DMA)
on the encoder/decoder shapes of the bench workload; also checks that both return identical bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomatching_amd import ops  # noqa: E402

S8 = 37171 * 8
SHAPES = [("enc fused N=640", S8, 640, 256, False, "f32"), ("enc out_proj", S8, 256, 256, False, "f32"),
          ("enc ffn1 relu->planes", S8, 1024, 256, True, "planes"), ("enc ffn2", S8, 256, 1024, False, "f32"),
          ("dec values N=1536", S8, 1536, 256, False, "f32"), ("res2 1x1 64->256", 890000, 256, 64, True, "f32"),
          ("small M=20000 N=768", 20000, 768, 256, False, "f32")]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main(rounds=3, iters=5):
    dev = "cuda"
    for name, M, N, K, relu, want in SHAPES:
        A = torch.randn(M, K, device=dev)
        W = ops.split_weight(torch.randn(N, K, device=dev))
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if N == 256 else None
        Ap = ops.split_rows(A)
        ref = ops.gemm(A, W, bias=b, R=R, relu=relu)
        got = ops.gemm_planes(Ap, W, bias=b, R=R, relu=relu, want=want)
        if want == "planes":
            same = torch.equal(got.float(), ref)
        else:
            same = torch.equal(got, ref)
        out = torch.empty(M, N, device=dev)
        outp = ops.new_planes(M, N, dev)
        t_old, t_new = [], []
        for _ in range(rounds):
            t_old.append(timeit(lambda: ops.gemm(A, W, bias=b, R=R, relu=relu, out=out), iters))
            t_new.append(timeit(lambda: ops.gemm_planes(Ap, W, bias=b, R=R, relu=relu, out=out, out_planes=outp,
                                                        want=want), iters))
        fl = 2.0 * M * N * K
        to, tn = min(t_old), min(t_new)
        print("%-24s M=%7d N=%5d K=%5d  old %8.1f us %6.1f TF | planes %8.1f us %6.1f TF  x%.2f  identical=%s"
              % (name, M, N, K, to * 1e6, fl / to / 1e12, tn * 1e6, fl / tn / 1e12, to / tn, same))
    x = torch.randn(S8, 256, device=dev)
    pl = ops.new_planes(S8, 256, dev)
    t = timeit(lambda: ops.split_rows(x, out=pl), 10)
    print("split_rows [%d,256]: %.1f us  (%.0f GB/s)" % (S8, t * 1e6, S8 * 256 * 10 / t / 1e9))


if __name__ == "__main__":
    main()
