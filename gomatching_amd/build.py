"""Build libgomatching_hip.so in-tree with hipcc for gfx950 (no torch extension machinery: the library
has a plain C ABI and links only the HIP runtime)."""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgomatching_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["abi.hip", "gemm_conv.hip", "gemm_bf16x6.hip", "gemm_f16x3.hip", "stem_pool.hip", "conv3x3_patch.hip", "ffn_fused.hip", "dec_tail.hip", "dec_tail2.hip", "proj_ln.hip", "dec_attn.hip", "dec_attn2.hip", "dec_inter.hip", "bneck_fused.hip", "bneck2.hip", "gemm_k256.hip", "gemm_small.hip", "msda.hip", "msda_any.hip", "norm.hip", "attn.hip", "elementwise.hip", "topk.hip",
           "detect.hip", "track.hip", "records.hip", "train.hip", "ingest.hip", "swin.hip", "vitae.hip", "attn_flash.hip", "tracker_rt.hip", "stream.hip", "lsa.cpp", "matcher_rt.cpp"]
FLAGS = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function"]
# device code is built without packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): DESIGN.md,
# "Tracker determinism"
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
NOPK_FILES = os.environ.get("GOM_NOPK_FILES")          # diagnostic: comma-separated subset; default = every .hip source


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stamp(path):
    h = hashlib.sha1()
    for dep in [path, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "tracker_tasks.h"), os.path.join(HERE, "..", "include", "gomatching_hip.h")]:
        with open(dep, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + _extra(os.path.basename(path))).encode())
    return h.hexdigest()


def _extra(src):
    if not src.endswith(".hip"):
        return []
    if NOPK_FILES is None or src in NOPK_FILES.split(","):
        return NOPK
    return []


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    stamp_file = obj + ".sha1"
    stamp = _stamp(path)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [_hipcc()] + FLAGS + _extra(src) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)
    if err.strip():                                          # (the x86 host pass does not know the device feature)
        sys.stderr.write(err + "\n")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return obj, True


def build(force=False, jobs=4):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
