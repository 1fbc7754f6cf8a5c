"""`python bench.py --gpus N` launches its own ranks (the reference starts its workers itself: train_net.py:198-209).
GPU: two ranks sharing cuda:0, records exchanged over gloo -- the whole N > 1 code path of bench.py except RCCL itself --
must print ONE line with n_gpus 2, 16 frames per step, both ranks seen in the gathered buffer, and the track ids of
the 16-frame clip identical to a single-rank run over the same 16 frames."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=1500):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.gpu
def test_self_launched_two_ranks_equal_single_rank():
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt-backends"]
    r2, l2 = _bench(["--gpus", "2"] + common, {"GOM_BENCH_BACKEND": "gloo"})
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert len(l2) == 1, r2.stdout[-2000:]
    two = json.loads(l2[0])
    assert two["n_gpus"] == 2 and two["config"]["frames_per_step"] == 16
    assert two["config"]["rccl_ranks_seen"] == 2 and two["config"]["emulated_world"] == 1
    assert two["config"]["collective_backend"] == "gloo"
    assert two["value"] > 0 and abs(two["value"] - 16 * 2 / (two["ms_per_step"] * 2e-3)) < 1e-6 * two["value"]
    r1, l1 = _bench(["--gpus", "1", "--frames-per-gpu", "16"] + common)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads(l1[0])
    ids2, ids1 = two["config"]["track_ids_per_frame"], one["config"]["track_ids_per_frame"]
    assert len(ids2) == len(ids1) == 16 and sum(len(x) for x in ids1) > 0
    assert ids2 == ids1
    assert two["config"]["tracks"] == one["config"]["tracks"]


def test_launcher_refuses_without_devices_and_propagates_failures():
    """No GPU here: with RCCL as the backend the launcher must refuse (rc 2) before starting anything; with the gloo dry-run
    backend the ranks start, fail at their first device call, and the launcher must return non-zero instead of hanging."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("needs a box without GPUs")
    r, lines = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=300)
    assert r.returncode == 2 and not lines and "only 0 GPU" in r.stderr
    r, lines = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"GOM_BENCH_BACKEND": "gloo"}, timeout=600)
    assert r.returncode != 0 and not lines
