"""Where a long-term match spends its time beside a saturated GPU: runs bench.py's pipelined loop with the tracker load of
W GPUs (--emulate-world W) and prints host preparation / launch / wait-for-device / assignment per match."""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gomatching_amd.modeling import meta_arch

prof = {"host_prep": 0.0, "issue": 0.0, "wait": 0.0, "matches": 0, "rows": 0, "assign": 0.0, "short": 0.0, "track_frames": 0.0}
G = meta_arch.GoMatching
_init = G.__init__


def init(self, *a, **k):
    _init(self, *a, **k)
    self._match_prof = prof


G.__init__ = init
for name, key in (("_assign", "assign"), ("precompute_short_term", "short"), ("track_frames", "track_frames")):
    real = getattr(G, name)

    def wrap(self, *a, _real=real, _key=key, **k):
        t0 = time.perf_counter()
        try:
            return _real(self, *a, **k)
        finally:
            prof[_key] += time.perf_counter() - t0
    setattr(G, name, wrap)

import bench
from gomatching_amd import ops as _ops0
_ops0.NATIVE_TRACKER = False          # these tools dissect the PYTHON loop of track_frames (the native runtime is one opaque call)
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--emulate-world", sys.argv[1] if len(sys.argv) > 1 else "8"]
bench.main()
m = max(prof["matches"], 1)
print("matches %d (avg %.0f selected rows); per match: host prep %.0f us, h2d + chain issue %.0f us, wait for device %.0f us; "
      "LSA+assign per call %.0f us" % (prof["matches"], prof["rows"] / m, prof["host_prep"] / m * 1e6, prof["issue"] / m * 1e6,
                                      prof["wait"] / m * 1e6, prof["assign"] / m * 1e6))
print("totals over the run (s): track_frames %.3f = matches %.3f + short-term precompute %.3f + assign %.3f + rest" % (
    prof["track_frames"], prof["host_prep"] + prof["issue"] + prof["wait"], prof["short"], prof["assign"]))
