"""Diagnostic build of the LDS-window MSDA kernel (csrc/msda.hip msda_window_kernel, single-buffer form): in-kernel cycles of a
workgroup's window geometry, owner part, and per level the fill issue, the wait + barriers and the samples.  Generated from the
product source (s_memtime stamps); the product kernel carries none.
    python tools/exp/msda_window_clock.py --build   (here)        python tools/exp/msda_window_clock.py   (GPU box)
STALE: the text anchors below match the kernel source before the ADDR template parameter (round 4); --build asserts until they are updated."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libmsda_window_clock.so")


def build():
    src = open(os.path.join(ROOT, "gomatching_amd", "csrc", "msda.hip")).read()

    def once(s, a, b):
        assert s.count(a) == 1, (a, s.count(a))
        return s.replace(a, b)
    src = once(src, 'template <int TY, int TX, int R, int CAP, int CAPB, int NB, bool DPP = false>\n__global__',
               '__device__ unsigned long long g_stamp[4096 * 16];\n#define NOW() __builtin_amdgcn_s_memtime()\n'
               'template <int TY, int TX, int R, int CAP, int CAPB, int NB, bool DPP = false>\n__global__')
    src = once(src, '    extern __shared__ __attribute__((aligned(16))) unsigned char win[];          // CAP lines of 128 bytes\n',
               '    extern __shared__ __attribute__((aligned(16))) unsigned char win[];\n    const unsigned long long t_start = NOW();\n'
               '    unsigned long long t_geo = 0, t_owner = 0, t_issue = 0, t_wait = 0, t_comp = 0, t_bar0 = 0;\n')
    src = once(src, '    // (the fill lambdas are defined below; level 0 of the double-buffered form is requested here, in front of the owner part)\n',
               '    t_geo = NOW() - t_start;\n')
    src = once(src, '    // ---- level by level: fill the window, then the level\'s four samples of every octet group ----\n',
               '    t_owner = NOW() - t_start - t_geo;\n')
    src = once(src, '        __syncthreads(); /* the previous level\'s reads are done */                                                \\\n'
                    '        fill(std::integral_constant<int, L>{}, win);                                                              \\\n'
                    '        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \\\n'
                    '        __syncthreads();                                                                                          \\\n'
                    '        level_samples(std::integral_constant<int, L>{}, win);                                                     \\\n',
               '        const unsigned long long a_ = NOW(); __syncthreads(); const unsigned long long b_ = NOW(); fill(std::integral_constant<int, L>{}, win); \\\n'
               '        const unsigned long long c_ = NOW(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); \\\n'
               '        const unsigned long long d_ = NOW(); level_samples(std::integral_constant<int, L>{}, win); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \\\n'
               '        const unsigned long long e_ = NOW(); t_bar0 += b_ - a_; t_issue += c_ - b_; t_wait += d_ - c_; t_comp += e_ - d_; \\\n')
    src = once(src, '    if (any_slow) {\n',
               '    if (blockIdx.x < 4096 && lane == 0) {\n        unsigned long long* o = g_stamp + (blockIdx.x * 4 + wave) % 4096 * 16;\n'
               '        o[0] = NOW() - t_start; o[1] = t_geo; o[2] = t_owner; o[3] = t_bar0; o[4] = t_issue; o[5] = t_wait; o[6] = t_comp;\n    }\n'
               '    if (any_slow) {\n')
    src += ('\nextern "C" int msda_window_clock_read(unsigned long long* host) {\n'
            '    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 4096 * 16);\n}\n')
    gen = os.path.join(HERE, "_msda_window_clock_gen.hip")
    open(gen, "w").write(src)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
                           "-packed-fp32-ops", "-I", os.path.join(ROOT, "gomatching_amd", "csrc"), "-I", os.path.join(ROOT, "include"), gen,
                           os.path.join(ROOT, "gomatching_amd", "csrc", "msda_any.hip"), "-o", SO])
    os.remove(gen)
    print("built", SO)


def main():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from gomatching_amd import ops
    from gomatching_amd.config import setup_cfg
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.predictor import GoMBatchPredictor
    from gomatching_amd.synth import make_clip
    from gomatching_amd.weights import synth_state_dict
    so = ctypes.CDLL(SO)
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    so.gom_msda_fused_forward_encoder.argtypes = [vp, ci, vp, vp, cl, ci, vp, vp, vp, ci, ci, ci, ci, vp]
    so.msda_window_clock_read.argtypes = [vp]
    B = 8
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.DEVICE = "cuda"
    clip = make_clip(B, 720, 1280, clip_id=0, num_rects=12)
    inputs, _ = GoMBatchPredictor(cfg, None).prepare([f[:, :, ::-1] for f in clip])
    model = GoMatching(cfg, synth_state_dict(cfg, seed=0), device="cuda", frames_per_step=B, use_graphs=False)
    x, _ = model.preprocess_image(inputs)
    feats = model.backbone.forward(x)
    det = model.detection_transformer
    src, geo = det.input_tokens([feats[k] for k in model.feature_names], B)
    S = geo["S"]
    rv = ops.linear(src, det.enc[0]["attn"]["raw_value"], R=geo["pos_w"][0], r_cols=384, r_period=S if geo["pos_periodic"] else 0)
    out = torch.empty((B * S, 256), device="cuda")
    val = rv[:, 384:]
    for _ in range(3):
        rc = so.gom_msda_fused_forward_encoder(rv.data_ptr(), rv.stride(0), geo["enc_ref"].data_ptr(), val.data_ptr(), S * 640, val.stride(0),
                                               geo["shapes"].data_ptr(), geo["lsi"].data_ptr(), out.data_ptr(), B, S, geo["hw0"][0],
                                               geo["hw0"][1], None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    host = np.zeros((4096, 16), np.uint64)
    assert so.msda_window_clock_read(host.ctypes.data) == 0
    h = host[host[:, 0] > 0].astype(np.float64)
    names = ["whole wave", "window geometry", "owner part (4 octet groups)", "barrier before a fill (x4)", "fill: DMA issue (x4)",
             "fill: wait + barrier (x4)", "samples of a level (x4)"]
    print("s_memtime ticks (100 MHz: 1 tick = 10 ns = ~21 cycles at 2.1 GHz), median over %d waves:" % len(h))
    for i, n in enumerate(names):
        print("  %-34s %8.0f ticks  = %7.2f us" % (n, np.median(h[:, i]), np.median(h[:, i]) * 0.01))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
