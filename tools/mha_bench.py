"""The decoder's two self-attentions (intra: 25 points of each of 800 queries; inter: 100 queries of each of 200 (frame, point)
pairs; 8 heads x 32) through gom_mha_core_f32 -- time per call and a hash of the output bits (for A/B of two builds of the
library: tools/ab_builds.sh swaps them)."""
import hashlib
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from gomatching_amd import ops

dev = "cuda"
B, nq, P, E = 8, 100, 25, 256
Q = B * nq * P
g = torch.Generator().manual_seed(0)
qk = torch.randn((Q, 2 * E), generator=g).to(dev)
v = torch.randn((Q, E), generator=g).to(dev)
qkv = torch.randn((Q, 3 * E), generator=g).to(dev)
attn = torch.empty((Q, E), device=dev)
ld = 3 * E
calls = {"intra (Lq = Lk = 25)": lambda: ops.mha_core(qk.view(-1), qk.view(-1)[E:], v, attn, B * nq, 1, 8, 32, P, P,
                                                     [P * 2 * E, 0, 2 * E, P * 2 * E, 0, 2 * E, P * E, 0, E, P * E, 0, E]),
         "inter (Lq = Lk = 100)": lambda: ops.mha_core(qkv.view(-1), qkv.view(-1)[E:], qkv.view(-1)[2 * E:], attn, B, P, 8, 32, nq, nq,
                                                      [nq * P * ld, ld, P * ld] * 3 + [nq * P * E, E, P * E])}
for name, fn in calls.items():
    attn.zero_()
    fn()
    torch.cuda.synchronize()
    digest = hashlib.sha1(attn.cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    print("%-24s %7.1f us per call   output sha1 %s" % (name, sorted(ts)[3], digest), flush=True)
