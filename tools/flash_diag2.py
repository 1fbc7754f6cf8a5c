import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops
DEV = "cuda"
N, heads, hd, qs = 64, 2, 128, 2.0
g = torch.Generator().manual_seed(N + hd)
B, C = 2, heads * hd
qkv = torch.randn(B * N, 3 * C, generator=g)
qkv[:, :C] *= qs
x = qkv.view(B, N, 3, heads, hd)
x[:, :, 2] = 0
for j in range(N):
    x[:, j, 2, :, j] = 1.0                      # v_j = e_j: the output row IS the probability row
q, k, v = qkv.double().view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
S = (q @ k.transpose(-2, -1)) * hd ** -0.5
P = S.softmax(-1)
got = ops.flash_attention(qkv.to(DEV), B, N, heads).cpu().view(B, N, heads, hd).permute(0, 2, 1, 3)[..., :N].double()
dlog = (got.clamp_min(1e-30).log() - P.log())
dlog = dlog - (P * dlog).sum(-1, keepdim=True)
bad = (dlog.abs() > 2e-5) & (P > 1e-6)
print("entries with |dlogp| > 2e-5:", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
from collections import Counter
print("by (b,h,query):", Counter([tuple(int(v) for v in i[:3]) for i in idx]).most_common(8))
print("by key:", sorted(Counter([int(i[3]) for i in idx]).items()))
bb, hh, qq = [int(v) for v in idx[0][:3]] if len(idx) else (1, 1, 1)
print("row", (bb, hh, qq), "dlogp by key:", ["%.0e" % float(v) for v in dlog[bb, hh, qq]])
# which d of q would explain dS_j = sum_d dq_d k_jd ?
dS = dlog[bb, hh, qq]
sol = torch.linalg.lstsq(k[bb, hh], dS[:, None]).solution[:, 0]         # 64 equations, 128 unknowns: min-norm
print("min-norm dq explaining it: top |dq| at d =", [(int(i), "%.1e" % float(sol[i])) for i in sol.abs().argsort(descending=True)[:10]])
print("q*scale at those d:", [("%.4f" % float(q[bb, hh, qq, i] * hd ** -0.5)) for i in sol.abs().argsort(descending=True)[:10]])
