"""The long-term match as ONE launch (csrc/match_fused.hip, gom_match_fused_f32) against the chain of 13 launches it replaces
(gom_match_scores_proj_f32; lstmatcher.py:333-381, transformer.py:60-96, gom_lstmatcher.py:429-445/510-547): every phase runs the
chain kernels' own wave tasks (csrc/tracker_tasks.h), so the trajectory scores must be EQUAL BIT FOR BIT -- whatever the grid
size, with and without the in-kernel descriptor upload, many times in a row on one barrier state; and the native tracker returns
the same ids with the fused path on and off."""
import numpy as np
import pytest
import torch

from helpers import mini_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _problem(heads, n_t, k, seed, with_decay=True):
    g = torch.Generator().manual_seed(seed)
    N, T, n_k = sum(n_t), len(n_t), n_t[k]
    d = heads.feature_dim
    pool = torch.randn(N + 9, d, generator=g).to(DEV)
    rows = torch.randperm(N + 9, generator=g)[:N].to(torch.int32)
    offs = np.concatenate([[0], np.cumsum(n_t)]).astype(np.int32)
    Np, lo = N - n_k, sum(n_t[:k])
    M = max(1, Np // 2)
    col_of = np.arange(Np) % M
    last = np.array([np.nonzero(col_of == m)[0].max() for m in range(M)])
    nonk = np.concatenate([np.arange(0, lo), np.arange(lo + n_k, N)])
    meta = np.concatenate([nonk, col_of, last, np.arange(lo, lo + n_k)]).astype(np.int32)
    xy = torch.rand(N, 2, generator=g) * 80
    boxes = torch.cat([xy, xy + 10 + torch.rand(N, 2, generator=g) * 30], 1)
    decay = (0.9 ** torch.arange(Np).float())
    # one descriptor block, as tracker_rt.hip lays it out: rows | offsets | meta | boxes | decay
    words = np.concatenate([rows.numpy(), offs, meta, boxes.numpy().reshape(-1).view(np.int32), decay.numpy().view(np.int32)])
    return dict(pool=pool, N=N, T=T, n_k=n_k, lo=lo, M=M, Np=Np, words=words, with_decay=with_decay)


def _views(dev_words, pr):
    N, T, Np, M, n_k = pr["N"], pr["T"], pr["Np"], pr["M"], pr["n_k"]
    o = np.cumsum([0, N, T + 1, 2 * Np + M + n_k, 4 * N, Np])
    rows, offs, meta = dev_words[o[0]:o[1]], dev_words[o[1]:o[2]], dev_words[o[2]:o[3]]
    boxes = dev_words[o[3]:o[4]].view(torch.float32)
    decay = dev_words[o[4]:o[5]].view(torch.float32) if pr["with_decay"] else None
    return rows, offs, meta, boxes, decay


def _heads(builtin):
    from gomatching_amd.modeling.roi_heads import build_roi_heads
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg(builtin, device=DEV)
    heads = build_roi_heads(cfg, synth_state_dict(cfg, seed=11), torch.device(DEV))
    return heads


def _proj(heads, pool):
    from gomatching_amd import ops
    m = heads._matcher(False)
    d = m.d
    proj = torch.empty((pool.shape[0], 4 * d), dtype=torch.float32, device=DEV)
    w_in, b_in = m.enc[0]["in"]
    w_q, b_q = m.dec[0]["in"]
    ops.gemm_small_rows(pool, w_in, b_in, proj[:, :3 * d])
    ops.gemm_small_rows(pool, w_q[:d], b_q[:d], proj[:, 3 * d:])
    return proj


def _run(heads, pr, proj, fused, upload=False, dev_words=None):
    from gomatching_amd import ops
    m = heads._matcher(False)
    desc = None
    if dev_words is None:
        dev_words = torch.from_numpy(pr["words"]).to(DEV)
    if upload:
        host = torch.from_numpy(pr["words"]).pin_memory()
        dev_words = torch.full_like(dev_words, -1)            # the kernel must fill it
        desc = (host, dev_words)
    rows, offs, meta, boxes, decay = _views(dev_words, pr)
    out = ops.match_scores_proj(pr["pool"], proj, rows, offs, meta, boxes, decay, pr["N"], pr["T"], pr["lo"], pr["lo"] + pr["n_k"],
                                pr["M"], m._enc_c, len(m.enc), m._dec_c, len(m.dec), m.d, m.heads, m.ffn, 128, 96, True, 50.0,
                                fused=fused, desc=desc)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("n_t,k", [([7, 0, 12, 5], 3), ([1, 1], 1), ([9, 9, 9, 9, 9, 9, 10], 6), ([3, 2, 1], 1), ([30, 34], 1), ([0, 2, 1], 2)])
@pytest.mark.parametrize("grid", [32, 5, 1])
def test_fused_match_equals_the_chain_bit_for_bit(n_t, k, grid):
    from gomatching_amd import ops
    heads = _heads("icdar15")
    m = heads._matcher(False)
    pr = _problem(heads, n_t, k, seed=sum(n_t) + k)
    assert ops._L().gom_match_fused_serves(pr["N"], pr["n_k"], len(m.enc), len(m.dec), m.d, m.heads, m.ffn, 1) == 1
    proj = _proj(heads, pr["pool"])
    chain = _run(heads, pr, proj, fused=False)
    assert chain.shape == (pr["n_k"], pr["M"]) and torch.isfinite(chain).all() and float(chain.abs().max()) > 0
    ops.check(ops._L().gom_match_fused_set_grid(grid), "gom_match_fused_set_grid")
    try:
        assert torch.equal(_run(heads, pr, proj, fused=True), chain)
        assert torch.equal(_run(heads, pr, proj, fused=True, upload=True), chain)
    finally:
        ops._L().gom_match_fused_set_grid(32)


def test_fused_match_without_decay_and_repeated_on_one_barrier_state():
    """No decay vector (short windows of the reference pass None) and 200 launches back to back: the barrier's arrival count
    returns to zero after every launch and the generation only moves forward, so one state serves a tracker's whole life."""
    from gomatching_amd import ops
    heads = _heads("icdar15")
    pr = _problem(heads, [6, 8, 7, 9], 3, seed=3, with_decay=False)
    proj = _proj(heads, pr["pool"])
    chain = _run(heads, pr, proj, fused=False)
    m = heads._matcher(False)
    dev_words = torch.from_numpy(pr["words"]).to(DEV)
    rows, offs, meta, boxes, decay = _views(dev_words, pr)
    L = ops._L()
    nws = L.gom_match_workspace_floats(pr["N"], pr["n_k"], m.d, m.ffn)
    ws = torch.empty((nws,), dtype=torch.float32, device=DEV)
    sync = torch.zeros((2,), dtype=torch.int32, device=DEV)
    status = torch.zeros((1,), dtype=torch.int32, device=DEV)
    outs = [torch.empty((pr["n_k"], pr["M"]), dtype=torch.float32, device=DEV) for _ in range(200)]
    for o in outs:
        ops.check(L.gom_match_fused_f32(pr["pool"].data_ptr(), pr["pool"].stride(0), proj.data_ptr(), proj.stride(0), rows.data_ptr(),
                                        offs.data_ptr(), meta.data_ptr(), boxes.data_ptr(), None, pr["N"], pr["T"], pr["lo"],
                                        pr["lo"] + pr["n_k"], pr["M"], m._enc_c, len(m.enc), m._dec_c, len(m.dec), m.d, m.heads, m.ffn,
                                        128.0, 96.0, 1, 50.0, ws.data_ptr(), nws, o.data_ptr(), sync.data_ptr(), status.data_ptr(),
                                        None, None, 0, ops._stream()), "gom_match_fused_f32")
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and int(sync[0].item()) == 0
    # barriers per launch: gather | per encoder layer attend, out, lin1, lin2 | per decoder layer kv, attend, out (+ lin1, lin2) | logits
    barriers = 1 + 4 * len(m.enc) + sum(3 + (2 if "lin1" in L else 0) for L in m.dec) + 1
    assert int(sync[1].item()) == 200 * barriers
    for o in outs:
        assert torch.equal(o, chain)


def test_fused_match_refuses_what_it_does_not_serve():
    from gomatching_amd import ops
    L = ops._L()
    assert L.gom_match_fused_serves(65, 3, 1, 1, 1024, 8, 1024, 1) == 0       # more rows than a wave has lanes (tiny attention)
    assert L.gom_match_fused_serves(40, 3, 1, 1, 1024, 8, 1024, 0) == 0       # no hoisted projections
    assert L.gom_match_fused_serves(40, 3, 1, 1, 256, 8, 1024, 1) == 0        # head_dim 32
    assert L.gom_match_fused_serves(40, 3, 1, 1, 1024, 8, 1024, 1) == 1
    assert L.gom_match_fused_set_grid(0) != 0 and L.gom_match_fused_set_grid(257) != 0


@pytest.mark.parametrize("builtin", ["icdar15", "pp_dstext"])
def test_native_tracker_ids_with_and_without_the_fused_match(builtin):
    """gom_tracker_run over a 106-frame synthetic trace (births, re-appearances, carried window): identical ids with the fused
    match on and off (the default) -- and the fused path really ran (GOM_TRACKER_SIZES-free check: the switch changes the launch
    count, not the results)."""
    from test_model_gpu import _synthetic_trace, _time_cost
    from gomatching_amd import ops
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.structures import Instances, Boxes
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg(builtin, device=DEV)
    sd = synth_state_dict(cfg, seed=7)
    size = (96, 128)

    def run(fused):
        ops.check(ops._L().gom_tracker_set_fused(1 if fused else 0), "gom_tracker_set_fused")
        try:
            model = GoMatching(cfg, sd, device=DEV)
            trace = _synthetic_trace(106, model.roi_heads.feature_dim, seed=5)
            dets = []
            for f, b in trace:
                inst = Instances(size)
                inst.reid_features = torch.from_numpy(f).to(DEV)
                inst.pred_boxes = Boxes(torch.from_numpy(b).to(DEV))
                dets.append(inst)
            it = iter(dets)
            model.detect_launch = lambda batched_inputs, time_cost: list(batched_inputs)
            model.detect_finish = lambda h, time_cost: [next(it) for _ in h]
            insts, id_count = model.batch_inference([{} for _ in range(100)], 0, 0, [], _time_cost())
            insts, id_count = model.batch_inference([{} for _ in range(6)], 1, id_count, insts, _time_cost())
            return [x.track_ids.cpu().tolist() for x in insts], int(id_count)
        finally:
            ops._L().gom_tracker_set_fused(0)

    a, b = run(True), run(False)
    assert a == b and max(max(f, default=0) for f in a[0]) > 5
