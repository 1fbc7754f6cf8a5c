"""Tracker determinism (DESIGN.md "Tracker determinism").

Round 1 shipped an open issue: the first evaluation of the tracker's scores differed from a recomputation in 11-50 % of
fresh processes whenever the detector of the next step ran on the other stream.  Root cause (round 2): the compiler had
turned the fmaf chains of the tracker's small GEMM into packed-fp32 `v_pk_fma_f32`, whose LOW half comes out wrong on
MI355X while waves of the bf16x6 GEMM kernel share the SIMD (tools/race_repro.py: 336-750 of 3000 launches wrong beside
that kernel, 0 beside the f16x3 / fp32 GEMMs, 0 with the kernel built without packed-fp32 instructions).  The library is
built without them (gomatching_amd/build.py); these tests pin that."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("load", ["bf16x6", "f16x3", "fp32"])
def test_small_gemm_beside_a_heavy_gemm(load):
    """The micro-reproducer: 3000 launches of the tracker's small GEMM on a high-priority stream beside a stream of heavy
    GEMMs of each back-end must all equal the idle-GPU result bit for bit."""
    r = _run(["tools/race_repro.py", load])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("load")][-1]
    assert ": 0 of 3000 launches differ" in line, line


@pytest.mark.parametrize("mode", ["bf16x6", "f16x3"])
def test_tracker_double_check_in_fresh_processes(mode):
    """Eight fresh child processes per back-end replay the clip that used to flip a track id, with every device result of the
    tracker computed twice (GOM_TRACKER_DOUBLE_CHECK=1) and the allocator pre-warmed -- the configuration that reproduced
    the issue in 40 of 40 processes before the fix (round 2; rounds 2-5 ran 21 children per back-end, 2 x 58 s of the suite): not
    one mismatch line, and the same ids in every process."""
    ref = os.path.join(ROOT, "gpurun_out", "flake_ref_%s.pt" % mode)
    if os.path.exists(ref):
        os.remove(ref)
    lines = []
    for i in range(8):                                        # the first child writes the reference ids
        r = _run(["tools/swin_flake.py", mode], env={"GOM_TRACKER_DOUBLE_CHECK": "1", "FLAKE_SWITCH": "prewarm_alloc"})
        assert r.returncode == 0, (i, r.stderr[-2000:])
        out = r.stdout + r.stderr
        lines += ["child %d: %s" % (i, l[:200]) for l in out.splitlines() if "MISMATCH" in l or l.startswith("DIFF")
                  or "not finite" in l]
        assert ("saved reference" in out) if i == 0 else ("SAME" in out), (i, out[-1500:])
    assert not lines, lines[:10]
