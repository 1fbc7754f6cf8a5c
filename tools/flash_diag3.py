import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from gomatching_amd import ops
DEV = "cuda"
N, heads, hd, qs = 64, 2, 128, 2.0
g = torch.Generator().manual_seed(N + hd)
B, C = 2, heads * hd
base = torch.randn(B * N, 3 * C, generator=g)
base[:, :C] *= qs


def run(qkv):
    q, k, v = qkv.double().view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = (((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ v).transpose(1, 2).reshape(B * N, C).float()
    got = ops.flash_attention(qkv.to(DEV), B, N, heads).cpu()
    e = (got - ref).abs().view(B, N, heads, hd)
    return float(e.max()), float(e[1, 1, 1].max())


print("as is:", run(base.clone()))
x = base.clone()
v = x.view(B, N, 3, heads, hd)
v[1, 1, 0, 1, 9] = torch.nextafter(v[1, 1, 0, 1, 9], torch.tensor(10.0))
print("element d=9 of the bad row moved by one ulp:", run(x))
# the same tie through the f16x3 GEMM: A = the bad q row times the scale (activations), W = identity
scale = np.float32(1.0) / np.sqrt(np.float32(128.0))
row = (base.view(B, N, 3, heads, hd)[1, 1, 0, 1].numpy() * scale).astype(np.float32)
A = torch.from_numpy(np.tile(row, (64, 1))).to(DEV)
W = ops.split_weight(torch.eye(128, device=DEV).contiguous(), kind="f16x3")
out = ops.gemm(A, W).cpu().numpy()
print("f16x3 GEMM, activation = that row, weight = I: max |out - in| = %.2e at d=%d" % (np.abs(out[0] - row).max(), int(np.abs(out[0] - row).argmax())))
# and as a WEIGHT operand (the weight split path scales rows by a power of two)
Wq = ops.split_weight(torch.from_numpy(np.tile(row, (64, 1))).to(DEV).contiguous(), kind="f16x3")
out2 = ops.gemm(torch.eye(128, device=DEV)[:64].contiguous(), Wq).cpu().numpy()
print("f16x3 GEMM, weight = that row: max err %.2e" % np.abs(out2[:, 0] - row[:64]).max())
