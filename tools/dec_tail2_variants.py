"""Schedule variants of csrc/dec_tail2.hip, built as stand-alone libraries from the same source with -D knobs and timed against
each other (and against form 1) in alternating bursts on one GPU at the decoder's shape.  WAVES=4|8 (environment, default 8): waves
per workgroup of the launches.
    python tools/dec_tail2_variants.py --build [name=flags ...]   (here: cross-compiles tools/exp/libdt2_<name>.so)
    python tools/dec_tail2_variants.py [M]                        (GPU box)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "exp")
SRC = os.path.join(ROOT, "gomatching_amd", "csrc", "dec_tail2.hip")
DEFAULT = {"base": "", "stamps": "-DT2_STAMPS"}


def build(variants):
    for name, flags in variants.items():
        out = os.path.join(EXP, "libdt2_%s.so" % name)
        cmd = ["hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
               "-shared", "-x", "hip", SRC, "-o", out] + flags.split()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            raise SystemExit(1)
        print("built", out, flags)


def main():
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gomatching_amd import ops
    from gomatching_amd.lib import SIGNATURES
    from test_dec_tail_gpu import _case
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    x, ffn, coord, qpos, ref, dim_t = _case(M, 1024, seed=3)
    dv = lambda t: t.to("cuda")
    g = torch.Generator().manual_seed(1)
    samp = dv(torch.randn((M, 256), generator=g))
    pw = (dv(torch.randn((256, 256), generator=g) / 16), dv(torch.randn((256,), generator=g) * 0.1), dv(1.0 + 0.2 * torch.randn((256,), generator=g)),
          dv(0.1 * torch.randn((256,), generator=g)))
    mk = lambda form, proj: ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos],
                                        dv(dim_t), proj_w=pw if proj else None, form=form)
    b1 = mk(1, True)
    blocks = {4: ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                             proj_w=pw, form=2, waves=4),
              8: ops.DecTail(tuple(dv(v) for v in ffn), [(dv(w), dv(b)) for w, b in coord], [(dv(w), dv(b)) for w, b in qpos], dv(dim_t),
                             proj_w=pw, form=2, waves=8)}
    WAVES = int(os.environ.get("WAVES", 8))
    b2 = blocks[WAVES]
    X, R = dv(x), dv(ref)
    want = ops.dec_tail(samp, b1, R, want_qpos=True, residual=X)
    libs = {}
    for path in sorted(glob.glob(os.path.join(EXP, "libdt2_*.so"))):
        lib = ctypes.CDLL(path)
        res, args = SIGNATURES["gom_dec_tail2_f32"]
        lib.gom_dec_tail2_f32.restype, lib.gom_dec_tail2_f32.argtypes = res, args
        libs[os.path.basename(path)[7:-3]] = lib
    out, nref, qp = torch.empty((M, 256), device="cuda"), torch.empty((M, 2), device="cuda"), torch.empty((M, 256), device="cuda")
    flag = torch.zeros((1,), dtype=torch.int32, device="cuda")
    p = ops._p

    def run(lib, blk=b2):
        pi, pb, pg, pbe = blk.proj
        rc = lib.gom_dec_tail2_f32(p(samp), 256, p(X), 256, p(blk.image), blk.wave_bytes[1], blk.F, p(pi), p(pb), p(pg), p(pbe), blk.eps,
                                   p(blk.inv1), p(blk.b1), p(blk.inv2), p(blk.b2), p(blk.gamma), p(blk.beta), blk.eps, p(blk.c_inv1),
                                   p(blk.c_b1), p(blk.c_inv2), p(blk.c_b2), p(blk.W3), p(blk.b3), p(R), p(blk.dim_t), p(blk.q_inv1),
                                   p(blk.q_b1), p(blk.q_inv2), p(blk.q_b2), p(out), 256, p(nref), p(qp), 256, M, blk.waves, p(flag), ops._stream())
        assert rc == 0, rc

    def burst(fn, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    for name, lib in libs.items():
        out.zero_()
        run(lib)
        torch.cuda.synchronize()
        print("%-14s max |d| vs form 1: tgt %.2e ref %.2e qpos %.2e, flag %d" % (name, float((out - want[0]).abs().max()),
              float((nref - want[1]).abs().max()), float((qp - want[2]).abs().max()), int(flag.item())))
    for name, lib in libs.items():
        if not hasattr(lib, "gom_dec_tail2_set_stamps"):
            continue
        try:
            lib.gom_dec_tail2_set_stamps
        except AttributeError:
            continue
        nwg = (M + 79) // 80
        st = torch.zeros((nwg, WAVES, 16), dtype=torch.int64, device="cuda")
        lib.gom_dec_tail2_set_stamps.argtypes = [ctypes.c_void_p]
        run(lib)
        torch.cuda.synchronize()
        assert lib.gom_dec_tail2_set_stamps(ctypes.c_void_p(st.data_ptr())) == 0
        run(lib)
        torch.cuda.synchronize()
        lib.gom_dec_tail2_set_stamps(None)
        t = st.double()
        names = ["prologue", "proj loop", "proj epilogue", "FFN loop", "FFN epilogue", "coord loop", "coord epilogue + sine", "qpos loop", "end"]
        med = t[:, :, :9].median(dim=0).values            # [wave][slot]
        print("%s: cycles since the wave's start (median over workgroups), per wave" % name)
        prev = torch.zeros(WAVES, dtype=torch.float64, device="cuda")
        for i, nm in enumerate(names):
            print("  %-24s %s   (+%s)" % (nm, ["%7.0f" % v for v in med[:, i].tolist()], ["%6.0f" % v for v in (med[:, i] - prev).tolist()]))
            prev = med[:, i]
        for i, nm in ((10, "GEMM1 (all 12 chunks)"), (11, "barrier + act + barrier"), (12, "GEMM2")):
            print("  %-24s %s" % (nm, ["%7.0f" % v for v in t[:, :, i].median(dim=0).values.tolist()]))
        print("  slowest workgroup's end: %.0f, fastest: %.0f" % (float(t[:, :, 8].max()), float(t[:, :, 8].min())))
    for rnd in range(3):
        line = "round %d  M = %d with out_proj: form 1 %.1f us" % (rnd, M, burst(lambda: ops.dec_tail(samp, b1, R, want_qpos=True, residual=X)))
        for name, lib in libs.items():
            line += " | %s %.1f" % (name, burst(lambda: run(lib)))
        print(line, flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        extra = dict(a.split("=", 1) for a in sys.argv[2:])
        build(extra or DEFAULT)
    else:
        main()
