"""Schedule variants of csrc/dec_attn2.hip, built as stand-alone libraries from the same source with -D knobs and timed against each
other in alternating bursts on one GPU at the decoder's shape (8 frames x 100 queries x 25 points); the `stamps` variant prints where a
wave's cycles go.
    python tools/dec_attn2_variants.py --build [name=flags ...]   (here: cross-compiles tools/exp/libda2_<name>.so)
    python tools/dec_attn2_variants.py                            (GPU box)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "exp")
SRC = os.path.join(ROOT, "gomatching_amd", "csrc", "dec_attn2.hip")
DEFAULT = {"base": "", "stamps": "-DA2_STAMPS"}


def build(variants):
    for name, flags in variants.items():
        out = os.path.join(EXP, "libda2_%s.so" % name)
        cmd = ["hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
               "-shared", "-x", "hip", SRC, "-o", out] + flags.split()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            raise SystemExit(1)
        print("built", out, flags)


def main():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from gomatching_amd import ops
    from gomatching_amd.lib import SIGNATURES
    dev = "cuda"
    B, nq, P = 8, 100, 25
    Q = B * nq * P
    g = torch.Generator().manual_seed(0)
    in_w = (torch.randn(768, 256, generator=g) / 16).to(dev)
    in_b = (torch.randn(768, generator=g) * 0.1).to(dev)
    out_w = (torch.randn(256, 256, generator=g) / 16).to(dev)
    out_b = (torch.randn(256, generator=g) * 0.1).to(dev)
    gamma, beta = (torch.rand(256, generator=g) + 0.5).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
    rw, rb = (torch.randn(384, 256, generator=g) / 16).to(dev), (torch.randn(384, generator=g) * 0.1).to(dev)
    x, pos = torch.randn(Q, 256, generator=g).to(dev), torch.randn(Q, 256, generator=g).to(dev)
    intra = ops.DecAttnBlock(in_w, in_b, out_w, out_b, gamma, beta, False, form=2)
    inter = ops.DecAttnBlock(in_w, in_b, out_w, out_b, gamma, beta, True, raw=(rw, rb), form=2)
    y, raw = torch.empty_like(x), torch.empty((Q, 384), device=dev)
    flag = torch.zeros((1,), dtype=torch.int32, device=dev)
    want_intra = ops.dec_attn(x, intra, B * nq, P, pos=pos).clone()
    want_inter, want_raw = [t.clone() for t in ops.dec_attn(x, inter, B * P, nq, inner=P, raw_pos=pos)]
    libs = {}
    for path in sorted(glob.glob(os.path.join(EXP, "libda2_*.so"))):
        lib = ctypes.CDLL(path)
        for fn in ("gom_dec_attn2_f32", "gom_dec_attn2_raw_f32"):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = SIGNATURES[fn]
        libs[os.path.basename(path)[7:-3]] = lib
    p = ops._p

    def run_intra(lib):
        assert lib.gom_dec_attn2_f32(p(x), 256, p(pos), 256, p(intra.image2), 1e-5, p(y), 256, B * nq, P, 1, 0, p(flag), None) == 0

    def run_inter(lib):
        assert lib.gom_dec_attn2_f32(p(x), 256, None, 0, p(inter.image2), 1e-5, p(y), 256, B * P, nq, P, 1, p(flag), None) == 0

    def run_raw(lib):
        assert lib.gom_dec_attn2_raw_f32(p(x), 256, p(inter.image2), 1e-5, p(y), 256, p(pos), 256, p(raw), 384, B * P, nq, P, p(flag), None) == 0

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    junk = torch.empty((768 << 20,), dtype=torch.uint8, device=dev)

    def timeit_cold(fn, n=12):
        """every launch behind a 768 MB fill: inputs and weights come from HBM, as between the layers of a step"""
        ts = []
        for _ in range(n):
            junk.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    for name, lib in libs.items():
        run_intra(lib)
        d0 = float((y - want_intra).abs().max())
        run_raw(lib)
        print("%-12s max |d| vs the product library: intra %.1e inter %.1e raw %.1e" % (name, d0, float((y - want_inter).abs().max()),
                                                                                      float((raw - want_raw).abs().max())))
    for rnd in range(3):
        print("round %d: " % rnd + " | ".join("%s intra %.1f inter %.1f inter+raw %.1f us" % (
            name, timeit(lambda: run_intra(lib)), timeit(lambda: run_inter(lib)), timeit(lambda: run_raw(lib))) for name, lib in libs.items()))
    print("cold (a 768 MB fill in front of every launch): " + " | ".join("%s intra %.1f inter %.1f inter+raw %.1f us" % (
        name, timeit_cold(lambda: run_intra(lib)), timeit_cold(lambda: run_inter(lib)), timeit_cold(lambda: run_raw(lib))) for name, lib in libs.items()))
    if "stamps" in libs:
        lib = libs["stamps"]
        lib.gom_dec_attn2_set_stamps.argtypes = [ctypes.c_void_p]
        names = ["prologue", "stage products", "stage epilogues", "waits + barriers", "attention", "row reload", "residual + LN", "total"]
        for label, fn, nwg in (("intra", run_intra, B * nq // 4), ("inter", run_inter, B * P), ("inter + raw", run_raw, B * P)):
            buf = torch.zeros((nwg * 8 * 8,), dtype=torch.int64, device=dev)
            assert lib.gom_dec_attn2_set_stamps(p(buf)) == 0
            fn(lib)
            fn(lib)
            torch.cuda.synchronize()
            t = buf.cpu().numpy().reshape(nwg, 8, 8).astype(np.float64)
            med = np.median(t.reshape(-1, 8), axis=0)
            print("%-12s cycles per wave (median over %d waves): " % (label, nwg * 8) + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, med)))
            w = np.median(t, axis=0)
            print("             total by wave index: " + " ".join("%.0f" % v for v in w[:, 7]) + " | products " + " ".join("%.0f" % v for v in w[:, 1])
                  + " | sync " + " ".join("%.0f" % v for v in w[:, 3]))
        lib.gom_dec_attn2_set_stamps(None)


if __name__ == "__main__":
    if "--build" in sys.argv:
        v = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a) or DEFAULT
        build(v)
    else:
        main()
