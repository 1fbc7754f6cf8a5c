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
q, k, v = qkv.double().view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
S = (q @ k.transpose(-2, -1)) * hd ** -0.5
P = S.softmax(-1)
ref = (P @ v).transpose(1, 2).reshape(B * N, C).float()
d = qkv.to(DEV)
a = ops.flash_attention(d, B, N, heads).cpu()
b2 = ops.flash_attention(d, B, N, heads).cpu()
print("deterministic:", bool(torch.equal(a, b2)))
err = (a - ref).abs().view(B, N, heads, hd)
rows = err.amax(-1)
bad = (rows > 5e-6).nonzero()
print("rows above 5e-6:", bad.tolist(), [float(rows[tuple(i)]) for i in bad])
for i in bad[:4]:
    bb, qq, hh = [int(x) for x in i]
    s = S[bb, hh, qq]
    qrow = (q[bb, hh, qq] * hd ** -0.5)
    print(" row", (bb, qq, hh), "S max %.3f min %.3f" % (float(s.max()), float(s.min())), "|qs| max %.3f min %.2e" % (float(qrow.abs().max()), float(qrow.abs().min())),
          "signed rel err of row: %s" % ["%.1e" % float(x) for x in ((a.view(B, N, heads, hd)[bb, qq, hh] - ref.view(B, N, heads, hd)[bb, qq, hh]) / ref.view(B, N, heads, hd)[bb, qq, hh].abs().clamp_min(1e-3))[:6]])
    # which single key, if its probability were off by a factor, explains the error best?
    diff = (a.view(B, N, heads, hd)[bb, qq, hh] - ref.view(B, N, heads, hd)[bb, qq, hh]).double()
    vv = v[bb, hh] - ref.view(B, N, heads, hd)[bb, qq, hh].double()[None]         # d out / d log p_j = p_j (v_j - out)
    coef = (vv @ diff) / (vv * vv).sum(-1)
    j = int(coef.abs().argmax())
    print("   best single-key explanation: key %d, p=%.4f, dlogp=%.2e, residual %.2e of %.2e" % (
        j, float(P[bb, hh, qq, j]), float(coef[j] / P[bb, hh, qq, j]), float((diff - coef[j] * vv[j]).abs().max()), float(diff.abs().max())))
    print("   k row |k| max %.3f; k0 overflow? %s" % (float(k[bb, hh, j].abs().max()), bool((k[bb, hh, j].abs() > 65504).any())))
bb, qq, hh = 1, 1, 1
out_ref = ref.view(B, N, heads, hd)[bb, qq, hh].double()
diff = (a.view(B, N, heads, hd)[bb, qq, hh].double() - out_ref)
A = (P[bb, hh, qq][:, None] * (v[bb, hh] - out_ref[None])).t()          # [hd, N]: d out / d log p_j
sol = torch.linalg.lstsq(A, diff[:, None]).solution[:, 0]
print("lstsq residual %.2e" % float((A @ sol - diff).abs().max()))
sol = sol - (P[bb, hh, qq] * sol).sum()
order = sol.abs().argsort(descending=True)[:12]
print("keys with largest dlogp:", [(int(j), "%.1e" % float(sol[j]), "p=%.4f" % float(P[bb, hh, qq, j])) for j in order])
# the same for a healthy row
qq2 = 2
out_ref2 = ref.view(B, N, heads, hd)[bb, qq2, hh].double()
diff2 = (a.view(B, N, heads, hd)[bb, qq2, hh].double() - out_ref2)
A2 = (P[bb, hh, qq2][:, None] * (v[bb, hh] - out_ref2[None])).t()
sol2 = torch.linalg.lstsq(A2, diff2[:, None]).solution[:, 0]
sol2 = sol2 - (P[bb, hh, qq2] * sol2).sum()
print("healthy row: max |dlogp| %.1e" % float(sol2.abs().max()))
print("all dlogp of the bad row by key:", ["%.0e" % float(x) for x in sol])
