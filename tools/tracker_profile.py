"""cProfile of the tracker's host side under the tracker load of W GPUs (bench.py --emulate-world W)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gomatching_amd.modeling import meta_arch
G = meta_arch.GoMatching
real = G.track_frames
prof = cProfile.Profile()


def wrapped(self, *a, **k):
    prof.enable()
    try:
        return real(self, *a, **k)
    finally:
        prof.disable()


G.track_frames = wrapped
import bench
from gomatching_amd import ops as _ops0
_ops0.NATIVE_TRACKER = False          # these tools dissect the PYTHON loop of track_frames (the native runtime is one opaque call)
sys.argv = ["bench.py", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--emulate-world", sys.argv[1] if len(sys.argv) > 1 else "8"]
bench.main()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
