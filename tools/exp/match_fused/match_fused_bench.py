"""One long-term match on an idle GPU: the chain of 13 launches (gom_match_scores_proj_f32) against the one-launch form
(gom_match_fused_f32, csrc/match_fused.hip) at several grid sizes.    python tools/match_fused_bench.py [N_per_frame ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_match_fused_gpu import _heads, _problem, _proj, _views, DEV   # noqa: E402
from gomatching_amd import ops                                          # noqa: E402


def main():
    per_frame = [int(a) for a in sys.argv[1:]] or [7]
    heads = _heads("icdar15")
    m = heads._matcher(False)
    L = ops._L()
    for n in per_frame:
        n_t = [n] * 7
        pr = _problem(heads, n_t, 6, seed=n)
        proj = _proj(heads, pr["pool"])
        dev_words = torch.from_numpy(pr["words"]).to(DEV)
        rows, offs, meta, boxes, decay = _views(dev_words, pr)
        nws = L.gom_match_workspace_floats(pr["N"], pr["n_k"], m.d, m.ffn)
        ws = torch.empty((nws,), dtype=torch.float32, device=DEV)
        traj = torch.empty((pr["n_k"], pr["M"]), dtype=torch.float32, device=DEV)
        sync = torch.zeros((2,), dtype=torch.int32, device=DEV)
        status = torch.zeros((1,), dtype=torch.int32, device=DEV)
        common = (pr["pool"].data_ptr(), pr["pool"].stride(0), proj.data_ptr(), proj.stride(0), rows.data_ptr(), offs.data_ptr(),
                  meta.data_ptr(), boxes.data_ptr(), decay.data_ptr(), pr["N"], pr["T"], pr["lo"], pr["lo"] + pr["n_k"], pr["M"],
                  m._enc_c, len(m.enc), m._dec_c, len(m.dec), m.d, m.heads, m.ffn, 128.0, 96.0, 1, 50.0, ws.data_ptr(), nws,
                  traj.data_ptr())

        def chain():
            ops.check(L.gom_match_scores_proj_f32(*common, ops._stream()), "chain")

        def fused():
            ops.check(L.gom_match_fused_f32(*common, sync.data_ptr(), status.data_ptr(), None, None, 0, ops._stream()), "fused")

        def timed(fn, reps=60):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps * 1e3

        line = "N = %3d rows (n_k %d): chain %7.1f us |" % (pr["N"], pr["n_k"], timed(chain))
        for grid in (8, 16, 32, 64, 128, 256):
            L.gom_match_fused_set_grid(grid)
            line += " G%d %.1f" % (grid, timed(fused))
        L.gom_match_fused_set_grid(32)
        assert int(status.item()) == 0
        print(line, flush=True)


if __name__ == "__main__":
    main()
