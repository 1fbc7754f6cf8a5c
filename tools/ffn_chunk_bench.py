"""Does running the row-local tail of an encoder layer (out_proj -> LN -> FFN1 -> FFN2 -> LN) chunk by chunk keep its
intermediates in the 256 MB Infinity Cache?  Same kernels, same arithmetic, only the launch order over row ranges changes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
M, d, ffn = 8 * 37171, 256, 1024
g = torch.Generator(device=dev).manual_seed(0)
samp = torch.randn(M, d, device=dev, generator=g)
src = torch.randn(M, d, device=dev, generator=g)
mk = lambda n, k: ops.prep_weight((torch.randn(n, k, device=dev, generator=g) / k ** 0.5).contiguous())
w_out, w1, w2 = mk(d, d), mk(ffn, d), mk(d, ffn)
b_out, b1, b2 = torch.zeros(d, device=dev), torch.zeros(ffn, device=dev), torch.zeros(d, device=dev)
gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
x1 = torch.empty(M, d, device=dev); x2 = torch.empty(M, d, device=dev); hid = torch.empty(M, ffn, device=dev)
y = torch.empty(M, d, device=dev); out = torch.empty(M, d, device=dev)


def tail(lo, hi):
    ops.gemm(samp[lo:hi], w_out, bias=b_out, R=src[lo:hi], out=x1[lo:hi])
    ops.layernorm(x1[lo:hi], gam, bet, out=x2[lo:hi])
    ops.gemm(x2[lo:hi], w1, bias=b1, relu=True, out=hid[lo:hi])
    ops.gemm(hid[lo:hi], w2, bias=b2, R=x2[lo:hi], out=y[lo:hi])
    ops.layernorm(y[lo:hi], gam, bet, out=out[lo:hi])


def run(chunk):
    for lo in range(0, M, chunk):
        tail(lo, min(M, lo + chunk))


ref = None
for chunk in (M, 131072, 65536, 32768, 16384, 8192, 4096):
    run(chunk); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run(chunk)
    gr.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    print("chunk %7d rows: %.3f ms per layer tail   (identical to unchunked: %s)" % (chunk, e0.elapsed_time(e1) / 5,
                                                                                 bool(torch.equal(out, ref))))
