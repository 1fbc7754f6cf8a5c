"""Tensor-level wrappers over the C ABI (torch = device memory + stream plumbing only).

Every function launches hand-written HIP kernels from libgomatching_hip.so on torch's current stream;
none has a CPU or torch-op fallback.
"""
import contextlib
import ctypes
import math

import numpy as np
import torch

from . import lib as _lib_mod
from .lib import check

_f32 = torch.float32


def _L():
    return _lib_mod.load()


# Optional launch profiler for bench.py's roofline leg: a list that receives (start_event, end_event, flops) for
# every launch of the dominant kernel (the 128x128-tile plain GEMM).  None = no events recorded.
_gemm_profile = None


_profile_scope = ""


@contextlib.contextmanager
def profile_scope(name):
    """Every profile record made inside the block carries `name` as its last element (bench.py: which launches belong to the
    decoder layers)."""
    global _profile_scope
    old, _profile_scope = _profile_scope, name
    try:
        yield
    finally:
        _profile_scope = old


def set_gemm_profile(collector):
    global _gemm_profile
    _gemm_profile = collector


def _p(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    # raw hipStream_t of torch's current stream; the private accessor costs ~0.3 us against ~8 us for
    # torch.cuda.current_stream().cuda_stream, which matters on the launch-bound tracker path
    return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or \
    (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if t.dtype != _f32 or not t.is_cuda or not t.is_contiguous():
            raise _lib_mod.GomError("expected a contiguous float32 CUDA tensor, got %s %s contiguous=%s" % (
                t.dtype, t.device, t.is_contiguous()))


# ------------------------------------------------------------------------------------------ GEMM
# "bf16x6": weights of the big contractions are pre-split into three bf16 planes and multiplied on the bf16
# matrix cores with fp32-level accuracy (csrc/gemm_bf16x6.hip); "fp32": exact-fp32 MFMA everywhere.
GEMM_MODE = "f16x3"


class SplitWeight:
    """Pre-split planes of an fp32 weight [N, K] (row slices keep the plane stride): kind "bf16x6" = [3, N, Kpad] bf16,
    kind "f16x3" = [2, N, Kpad] fp16 of the power-of-two-scaled rows + `inv_scale` [N] fp32."""

    def __init__(self, planes, N, K, conv_shape=None, kind="bf16x6", inv_scale=None):
        self.planes, self.N, self.K, self.conv_shape, self.kind, self.inv_scale = planes, N, K, conv_shape, kind, inv_scale

    @property
    def shape(self):
        return (self.N, self.K)

    def __getitem__(self, sl):
        assert isinstance(sl, slice) and sl.step in (None, 1)
        a, b, _ = sl.indices(self.N)
        return SplitWeight(self.planes[:, a:b], b - a, self.K, kind=self.kind,
                           inv_scale=None if self.inv_scale is None else self.inv_scale[a:b])


def split_weight(w, conv_shape=None, kind=None):
    """w: fp32 [N, K] on the GPU (rows may be strided: a column slice of a wider buffer, as the K / V operands of an
    attention product are) -> SplitWeight of the current GEMM_MODE's kind."""
    assert w.dim() == 2 and w.stride(1) == 1 and w.dtype == _f32 and w.is_cuda
    kind = kind or (GEMM_MODE if GEMM_MODE in ("bf16x6", "f16x3") else "bf16x6")
    N, K = w.shape
    ldw = w.stride(0) if N > 1 else K
    Kpad = (K + 31) // 32 * 32
    if kind == "f16x3":
        planes = torch.empty((2, N, Kpad), dtype=torch.float16, device=w.device)
        inv = torch.empty((N,), dtype=_f32, device=w.device)
        check(_L().gom_split_f16x2(_p(w), ldw, N, K, _p(planes), Kpad, _p(inv), _stream()), "gom_split_f16x2")
        return SplitWeight(planes, N, K, conv_shape, "f16x3", inv)
    planes = torch.empty((3, N, Kpad), dtype=torch.bfloat16, device=w.device)
    check(_L().gom_split_bf16x3(_p(w), ldw, N, K, _p(planes), Kpad, _stream()), "gom_split_bf16x3")
    return SplitWeight(planes, N, K, conv_shape)


_range_flags = {}


def _switch(name, default=True):
    """Build-time switches of the f16x3 path; the environment (GOM_<NAME>=0/1) overrides the default for same-box A/B runs."""
    import os
    v = os.environ.get("GOM_" + name)
    return default if v is None else v not in ("0", "false", "False", "")


def _dev_index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def range_flag(device):
    """Device word the f16x3 kernels set when a result is not finite (an activation beyond fp16's range)."""
    key = _dev_index(device)
    if key not in _range_flags:
        _range_flags[key] = torch.zeros((1,), dtype=torch.int32, device=torch.device("cuda", key))
    return _range_flags[key]


def check_range_flag(device):
    """Host check at a sync point: raises instead of letting an out-of-range activation pass as a result."""
    key = _dev_index(device)
    words = [f for f in [_range_flags.get(key)] if f is not None]
    up = [f for f in words if int(f.item()) != 0]
    if up:
        for f in up:
            f.zero_()
        raise _lib_mod.GomError("f16x3 GEMM produced a non-finite value: an activation left fp16's range (|x| > 65504) "
                                "or the input was not finite; run with ops.GEMM_MODE = 'bf16x6'")


@contextlib.contextmanager
def gemm_mode(mode):
    """Run a block (model construction and / or a forward pass) under another contraction back-end; the model's precision
    fallback builds and runs its bf16x6 twin of the detector this way (GoMatching._fallback_detect)."""
    global GEMM_MODE
    old, GEMM_MODE = GEMM_MODE, mode
    try:
        yield
    finally:
        GEMM_MODE = old


def flag_nonfinite(x, flag):
    """flag |= 1 on the device when x holds an Inf / NaN (result check of the back-ends whose GEMMs carry no range flag)."""
    _chk_f32(x)
    check(_L().gom_flag_nonfinite_f32(_p(x), x.numel(), _p(flag), _stream()), "gom_flag_nonfinite_f32")


def prep_weight(w, min_n=33):
    """Weight preparation policy for the detector's nn.Linear weights."""
    if GEMM_MODE in ("bf16x6", "f16x3") and w.shape[0] >= min_n:
        return split_weight(w.contiguous())
    return w


def prep_conv_weight(w_ohwi):
    if GEMM_MODE in ("bf16x6", "f16x3") and w_ohwi.shape[0] >= 33:
        Cout = w_ohwi.shape[0]
        return split_weight(w_ohwi.reshape(Cout, -1).contiguous(), conv_shape=tuple(w_ohwi.shape))
    return w_ohwi


def _gemm_split(A, W, bias, scale, A2, rows, R, relu, out, M, r_cols=None, r_period=0):
    if A2 is not None:                                       # the bf16x6 kernel has one A operand: add first
        assert rows is None
        A = add(A.contiguous(), A2.contiguous())
    K, N = A.shape[1], W.N
    assert W.K == K
    lda = A.stride(0) if A.shape[0] > 1 else K
    if out is None:
        out = torch.empty((M, N), dtype=_f32, device=A.device)
    ldc = out.stride(0) if out.shape[0] > 1 else N
    ldr = (R.stride(0) if R.shape[0] > 1 else R.shape[1]) if R is not None else 0
    pl = W.planes
    prof = _gemm_profile if (_gemm_profile is not None and M > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = (r_cols if r_cols is not None else N) if R is not None else 0
    if W.kind == "f16x3":
        check(_L().gom_gemm_f32_f16x3_rp(_p(A), _p(rows), lda, _p(pl), pl.stride(0), pl.stride(1), _p(W.inv_scale),
                                         _p(scale), _p(bias), _p(R), ldr, rc, int(r_period),
                                         2 if relu == "gelu" else (1 if relu else 0), _p(out), ldc, M, N, K,
                                         _p(range_flag(A.device)), _stream()), "gom_gemm_f32_f16x3_rp")
    else:
        assert not r_period, "a periodic residual is served by the f16x3 kernel only"
        assert relu in (False, True, 0, 1), "only the f16x3 kernel has a GELU epilogue (use gemm_gelu)"
        check(_L().gom_gemm_f32_bf16x6(_p(A), _p(rows), lda, _p(pl), pl.stride(0), pl.stride(1), _p(scale),
                                       _p(bias), _p(R), ldr, rc, 1 if relu else 0, _p(out), ldc, M, N, K, _stream()),
              "gom_gemm_f32_bf16x6")
    if prof is not None:
        e1.record()
        nbytes = 4.0 * M * K + 2.0 * pl.shape[0] * N * pl.shape[2] + 4.0 * M * N \
            + (4.0 * (r_period or M) * rc if R is not None else 0.0)
        prof.append((e0, e1, 2.0 * M * N * K, nbytes, "%dx%dx%d" % (M, N, K), _profile_scope))
    return out


SPLITK_MAX_ROWS = 1024
SMALL_GEMM_OUTPUTS = 1 << 16         # M*N up to which gemm(..., small=True) runs the patch-per-wave VALU kernel (tracker logits)
SMALL_GEMM_ROWS = 64                 # ... and any N up to this many rows (csrc/matcher_rt.cpp `linear` applies the same rule)


def gemm(A, W, bias=None, scale=None, A2=None, rows=None, R=None, relu=False, out=None, M=None, splitk=None,
         r_cols=None, small=False, r_period=0):
    """C = act((A[+A2])[M,K] @ W[N,K]^T * scale + bias + R).  A may be a 2-D row-strided view
    (stride(1) == 1); W likewise (row slices of a weight matrix)."""
    if isinstance(W, SplitWeight):
        assert A.dim() == 2 and A.stride(1) == 1
        if M is None:
            M = A.shape[0] if rows is None else rows.numel()
        if A2 is not None:
            assert A2.shape == A.shape and A2.stride() == A.stride()
        return _gemm_split(A, W, bias, scale, A2, rows, R, relu, out, M, r_cols, r_period)
    assert not r_period, "a periodic residual (r_period) is served by the f16x3 split-weight kernel only"
    assert A.dim() == 2 and W.dim() == 2 and A.stride(1) == 1 and W.stride(1) == 1
    if R is not None and r_cols is not None and r_cols < W.shape[0]:
        # exact-fp32 kernel has no column-limited residual: run the two column blocks as two launches
        Mrows = A.shape[0] if rows is None else rows.numel()
        if out is None:
            out = torch.empty((Mrows, W.shape[0]), dtype=_f32, device=A.device)
        gemm(A, W[:r_cols], bias=None if bias is None else bias[:r_cols], scale=scale, A2=A2, rows=rows, R=R,
             relu=relu, out=out[:, :r_cols], M=M)
        gemm(A, W[r_cols:], bias=None if bias is None else bias[r_cols:], scale=scale, A2=A2, rows=rows, relu=relu,
             out=out[:, r_cols:], M=M)
        return out
    K = A.shape[1]
    assert W.shape[1] == K
    N = W.shape[0]
    if M is None:
        M = A.shape[0] if rows is None else rows.numel()
    lda, ldw = A.stride(0) if A.shape[0] > 1 else K, W.stride(0) if N > 1 else K
    if A2 is not None:
        assert A2.shape == A.shape and A2.stride() == A.stride()
    if out is None:
        out = torch.empty((M, N), dtype=_f32, device=A.device)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= M and out.shape[1] == N
    ldc = out.stride(0) if out.shape[0] > 1 else N
    ldr = 0
    if R is not None:
        assert R.dim() == 2 and R.stride(1) == 1 and R.shape[1] == N
        ldr = R.stride(0) if R.shape[0] > 1 else N
    if rows is not None:
        assert rows.dtype == torch.int32 and rows.is_contiguous()
    # `small` marks the tracker's latency-bound products (kernel choice must never depend on how many frames share a
    # step, so this is an explicit request): up to SMALL_GEMM_OUTPUTS outputs one wave owns a column for 8 rows on the
    # VALU (7-20 us); larger ones take the deterministic split-K (20-50 us) instead of looping the whole K in few
    # workgroups (85-95 us whatever M is; tools/skinny_bench.py)
    if small and A2 is None and M > 0:
        if (M <= SMALL_GEMM_ROWS or M * N <= SMALL_GEMM_OUTPUTS) and K % 4 == 0 and lda % 4 == 0 and ldw % 4 == 0:
            check(_L().gom_gemm_small_f32(_p(A), _p(rows), lda, _p(W), ldw, _p(scale), _p(bias), _p(R), ldr,
                                          1 if relu else 0, _p(out), ldc, M, N, K, _stream()), "gom_gemm_small_f32")
            return out
        splitk = M <= SPLITK_MAX_ROWS                        # beyond that the partial sums cost more than they save
    if splitk is None:
        splitk = 0 < M <= 128 and N * K >= (1 << 18) and K >= 256      # skinny, weight-read bound
    if splitk and A2 is None and M > 0:
        nbytes = _L().gom_gemm_splitk_workspace_bytes(M, N, K)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=A.device)
        check(_L().gom_gemm_f32_splitk(_p(A), _p(rows), lda, _p(W), ldw, _p(scale), _p(bias), _p(R), ldr,
                                       1 if relu else 0, _p(out), ldc, M, N, K, _p(ws), nbytes, _stream()),
              "gom_gemm_f32_splitk")
        return out
    prof = _gemm_profile if (_gemm_profile is not None and N > 64 and M > 0 and GEMM_MODE == "fp32") else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().gom_gemm_f32(_p(A), _p(A2), _p(rows), lda, _p(W), ldw, _p(scale), _p(bias), _p(R), ldr,
                            1 if relu else 0, _p(out), ldc, M, N, K, _stream()), "gom_gemm_f32")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N + (M * N if R is not None else 0)),
                     "%dx%dx%d" % (M, N, K), _profile_scope))
    return out


# f16x3 back-end: pointwise convolutions with 256 input channels (res4 conv3) on the row-resident K = 256 kernel.  MEASURED in the
# step (bench.py, one box): 157 us per launch against the tile kernel's 150 -- the shortcut's residual reads and 16-byte stores
# cost what the missing A staging saves at 441 one-per-CU tiles.  OFF; GOM_PW_K256=1 for A/B runs.
PW_K256 = _switch("PW_K256", default=False)
PW_K256_MIN_ROWS = 16384
CONV3_PATCH = _switch("CONV3_PATCH")   # f16x3 back-end: 3x3 / stride 1 convolutions on the patch-resident kernel


def conv2d_nhwc(x, w_ohwi, scale=None, shift=None, R=None, relu=False, stride=1, pad=0):
    """x [B,H,W,Cin] -> [B,OH,OW,Cout]; w [Cout,KH,KW,Cin] fp32, or a SplitWeight made by prep_conv_weight."""
    split = isinstance(w_ohwi, SplitWeight)
    _chk_f32(x, None if split else w_ohwi, scale, shift, R)
    B, H, Wd, Cin = x.shape
    Cout, KH, KW, Cin2 = w_ohwi.conv_shape if split else w_ohwi.shape
    assert Cin == Cin2
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (Wd + 2 * pad - KW) // stride + 1
    y = torch.empty((B, OH, OW, Cout), dtype=_f32, device=x.device)
    if R is not None:
        assert R.shape == y.shape
    if split:
        pl = w_ohwi.planes
        M = B * OH * OW
        splits = _L().gom_conv_bf16x6_splits(M, Cout, KH * KW * Cin)     # few tiles x long K (input_proj[3]): slice K
        ws, nbytes = None, 0
        if splits > 1:
            nbytes = 4 * splits * M * Cout
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        if (w_ohwi.kind == "f16x3" and CONV3_PATCH and KH == 3 and KW == 3 and stride == 1 and pad == 1 and R is None and splits <= 1
                and _L().gom_conv3x3_patch_supported(Cin, Cout)):
            # the bottlenecks' 3x3 / 1 convolutions: input patch resident in LDS (csrc/conv3x3_patch.hip)
            img = getattr(w_ohwi, "patch_image", None)
            if img is None:                                  # fragment-linear image of the planes, built once per weight
                nb = _L().gom_conv3x3_patch_image_bytes(Cin, Cout)
                img = torch.empty((nb,), dtype=torch.uint8, device=pl.device)
                check(_L().gom_conv3x3_patch_image(_p(pl), pl.stride(0), pl.stride(1), Cin, Cout, _p(img), nb, _stream()),
                      "gom_conv3x3_patch_image")
                w_ohwi.patch_image = img
            prof = _gemm_profile
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(_L().gom_conv3x3_patch_f32_f16x3(_p(x), _p(img), _p(w_ohwi.inv_scale), _p(scale), _p(shift), 1 if relu else 0,
                                                   _p(y), B, H, Wd, Cin, Cout, _p(range_flag(x.device)), _stream()),
                  "gom_conv3x3_patch_f32_f16x3")
            if prof is not None:
                e1.record()
                prof.append((e0, e1, 2.0 * M * Cout * 9 * Cin, 4.0 * M * (Cin + Cout) + 4.0 * 9 * Cin * Cout,
                             "conv3:%dx%dx%d" % (M, Cout, 9 * Cin), _profile_scope))
            return y
        if (w_ohwi.kind == "f16x3" and PW_K256 and KH == 1 and KW == 1 and stride == 1 and pad == 0 and Cin == 256 and Cout % 32 == 0
                and Cout >= 512 and M >= PW_K256_MIN_ROWS and splits <= 1 and x.is_contiguous()):
            # conv3 of the res4 bottlenecks (256 -> 1024, + BN + shortcut + ReLU): K = 256 is the row-resident kernel's shape
            # (csrc/gemm_k256.hip: rows as fragments once, weights by LDS-DMA, no A staging) -- FrozenBN's scale folds into the
            # image's inverse row scale, its shift is the bias
            key = (id(scale), id(shift))
            cache = w_ohwi.__dict__.setdefault("k256_images", {})
            img = cache.get(key)
            if img is None:
                inv = w_ohwi.inv_scale if scale is None else (w_ohwi.inv_scale * scale).contiguous()
                nb = _L().gom_gemm_k256_image_bytes(Cout, Cin)
                img = torch.empty((nb,), dtype=torch.uint8, device=pl.device)
                check(_L().gom_gemm_k256_image(_p(pl), pl.stride(0), pl.stride(1), _p(inv), _p(shift), Cout, Cin, _p(img), nb,
                                               _stream()), "gom_gemm_k256_image")
                cache[key] = img = (img, inv, scale, shift)         # (the vectors stay referenced: ids are the cache key)
            prof = _gemm_profile
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(_L().gom_gemm_k256_rp_f32(_p(x), None, Cin, _p(img[0]), _p(R), Cout if R is not None else 0,
                                            Cout if R is not None else 0, 0, 1 if relu else 0, _p(y), Cout, M, Cout, Cin, 1,
                                            _p(range_flag(x.device)), _stream()), "gom_gemm_k256_rp_f32")
            if prof is not None:
                e1.record()
                prof.append((e0, e1, 2.0 * M * Cout * Cin, 4.0 * M * Cin + 4.0 * M * Cout * (2 if R is not None else 1) + 4.0 * Cout * Cin,
                             "pwk256:%dx%dx%d" % (M, Cout, Cin), _profile_scope))
            return y
        if w_ohwi.kind == "f16x3":
            # a pointwise convolution IS a launch of the GEMM tile kernel (dispatch<0, 0> in csrc/gemm_f16x3.hip): bench.py's
            # roofline sample of that kernel covers these launches too (label "pw:")
            prof = _gemm_profile if (_gemm_profile is not None and KH == 1 and stride == 1 and pad == 0 and splits <= 1
                                     and Cout > 64) else None
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _after = lambda: (e1.record(), prof.append((e0, e1, 2.0 * M * Cout * Cin, 4.0 * M * Cin + 4.0 * M * Cout * (
                    2 if R is not None else 1) + 4.0 * Cout * Cin, "pw:%dx%dx%d" % (M, Cout, Cin), _profile_scope)))
            check(_L().gom_conv2d_nhwc_f32_f16x3(_p(x), _p(pl), pl.stride(0), pl.stride(1), _p(w_ohwi.inv_scale),
                                                 _p(scale), _p(shift), _p(R), 1 if relu else 0, _p(y), B, H, Wd, Cin,
                                                 Cout, KH, KW, stride, pad, _p(ws), nbytes, splits,
                                                 _p(range_flag(x.device)), _stream()), "gom_conv2d_nhwc_f32_f16x3")
            if prof is not None:
                _after()
            return y
        check(_L().gom_conv2d_nhwc_f32_bf16x6_splitk(_p(x), _p(pl), pl.stride(0), pl.stride(1), _p(scale), _p(shift),
                                                     _p(R), 1 if relu else 0, _p(y), B, H, Wd, Cin, Cout, KH, KW, stride,
                                                     pad, _p(ws), nbytes, splits, _stream()),
              "gom_conv2d_nhwc_f32_bf16x6_splitk")
        return y
    check(_L().gom_conv2d_nhwc_f32(_p(x), _p(w_ohwi), _p(scale), _p(shift), _p(R), 1 if relu else 0, _p(y), B, H, Wd,
                                   Cin, Cout, KH, KW, stride, pad, _stream()), "gom_conv2d_nhwc_f32")
    return y


# ------------------------------------------------------------------------------------------ norms
def layernorm(x, gamma, beta, residual=None, eps=1e-5, out=None):
    _chk_f32(x, gamma, beta, residual)
    D = x.shape[-1]
    rows = x.numel() // D
    if out is None:
        out = torch.empty_like(x)
    check(_L().gom_layernorm_f32(_p(x), _p(residual), _p(gamma), _p(beta), _p(out), rows, D, eps, _stream()),
          "gom_layernorm_f32")
    return out


def groupnorm32_into(x, gamma, beta, out_view, out_batch_stride, eps=1e-5):
    """x [B,HW,256] -> out_view (pointer to out[0, offset, 0] of a [B,S,256] buffer)."""
    _chk_f32(x, gamma, beta)
    B, HW, C = x.shape
    ws = torch.empty((B * 64,), dtype=torch.float64, device=x.device)
    check(_L().gom_groupnorm32_nhwc_f32(_p(x), _p(gamma), _p(beta), _p(ws), _p(out_view), out_batch_stride, B, HW, C,
                                        eps, _stream()), "gom_groupnorm32_nhwc_f32")


_masked_streams = {}


def masked_stream(mask_words, device):
    """A torch stream (ExternalStream over gom_stream_create_cu_mask) restricted to the CUs of `mask_words` (list of uint32).
    One hardware queue per (device, mask) for the life of the process: a re-reservation with the same split gets the same queue
    back, so repeated reservations do not accumulate queues.  The queues are deliberately never destroyed: torch's caching
    allocator remembers every stream a block was allocated or recorded on and records an event there when the block is freed --
    possibly long after a reservation ended (gom_stream_destroy under it aborts the process in HIPEvent::record)."""
    key = (_dev_index(device), tuple(int(w) & 0xFFFFFFFF for w in mask_words))
    st = _masked_streams.get(key)
    if st is None:
        arr = (ctypes.c_uint32 * len(mask_words))(*key[1])
        out = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(_L().gom_stream_create_cu_mask(arr, len(mask_words), ctypes.byref(out)), "gom_stream_create_cu_mask")
        st = torch.cuda.ExternalStream(out.value, device=device)
        _masked_streams[key] = st
    return st


K256_GEMM = _switch("K256_GEMM")   # f16x3 back-end: K = 256 products on the row-resident kernel where it measures faster (below)
K256_MAX_ROWS = 1 << 16  # "short" problems (the decoder's Q side: M = frames x queries x points)
K256_LONG = _switch("K256_LONG")   # long problems: N >= 512 on the kernel's whole-line-store form (off: N >= 1024, 16-byte stores)


def k256_wins(M, N, has_a2):
    """Kernel choice from tools/gemm_k256_bench.py (interleaved A/B on MI355X).  Both kernels return the same
    bits, so this is a pure speed rule.  Short problems: N = 256 (21 vs 24 us at M = 20 000) and everything with a second
    addend (the tile kernel needs an `add` launch first: 23-54 vs 32-56 us); long problems (whole-line-store form of the kernel):
    N >= 512 (at M = 297 368: N = 640 with the periodic position table 354 vs 449 us, N = 1536 747 vs 944 us; the tile kernel
    keeps N = 256: 170 vs 175 us)."""
    if M <= K256_MAX_ROWS:
        return N == 256 or has_a2
    return N >= (512 if K256_LONG else 1024)


class K256Linear:
    """An nn.Linear with in_features 256 prepared for gom_gemm_k256_f32: fragment-linear image of the f16x3 planes + inverse
    row scales + bias (csrc/gemm_k256.hip).  `W` (SplitWeight, possibly a row slice) and `bias` stay available for the tile
    kernel, which serves long problems."""

    def __init__(self, W, bias=None):
        assert isinstance(W, SplitWeight) and W.kind == "f16x3"
        nbytes = _L().gom_gemm_k256_image_bytes(W.N, W.K)
        if nbytes < 0:
            raise _lib_mod.GomError("row-resident GEMM kernel does not serve N %d / K %d" % (W.N, W.K))
        pl = W.planes
        self.image = torch.empty((nbytes,), dtype=torch.uint8, device=pl.device)
        check(_L().gom_gemm_k256_image(_p(pl), pl.stride(0), pl.stride(1), _p(W.inv_scale), _p(bias), W.N, W.K,
                                       _p(self.image), nbytes, _stream()), "gom_gemm_k256_image")
        self.W, self.bias, self.N, self.K = W, bias, W.N, W.K


def k256_linear(w, bias=None):
    """(weight, bias) of a K = 256 layer -> K256Linear when the back-end and shape allow, else the pair for ops.gemm."""
    if K256_GEMM and GEMM_MODE == "f16x3" and isinstance(w, SplitWeight) and w.kind == "f16x3" and w.K == 256 and w.N % 32 == 0:
        return K256Linear(w, bias)
    return (w, bias)


def linear(x, lin, A2=None, R=None, relu=False, r_cols=None, out=None, groups=0, r_period=0):
    """act((x [+ A2]) @ W^T + b [+ R]) for `lin` = K256Linear or a (weight, bias) pair; r_period > 0: row m adds
    R[m % r_period] (ops.gemm).  `groups`: column groups of the row-resident kernel's launch; the product always launches ONE
    (every workgroup splits its rows once and walks all columns): splitting the columns over more workgroups re-splits the
    rows per group and measured slower in the step (the decoder's 30 launches: 689 -> 878 us, bench.py same box)."""
    if not isinstance(lin, K256Linear):
        return gemm(x, lin[0], bias=lin[1], A2=A2, R=R, relu=relu, r_cols=r_cols, out=out, r_period=r_period)
    M = x.shape[0]
    # an explicit `groups` forces the kernel (tests, tools)
    if M == 0 or not (groups or k256_wins(M, lin.N, A2 is not None)):
        return gemm(x, lin.W, bias=lin.bias, A2=A2, R=R, relu=relu, r_cols=r_cols, out=out, r_period=r_period)
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == lin.K and x.dtype == _f32
    lda = x.stride(0) if M > 1 else lin.K
    if A2 is not None:
        assert A2.shape == x.shape and A2.stride() == x.stride() and A2.dtype == _f32
    N = lin.N
    if out is None:
        out = torch.empty((M, N), dtype=_f32, device=x.device)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= M and out.shape[1] == N
    ldr, rc = 0, 0
    if R is not None:
        assert R.dim() == 2 and R.stride(1) == 1 and R.dtype == _f32
        ldr, rc = (R.stride(0) if R.shape[0] > 1 else R.shape[1]), (r_cols if r_cols is not None else N)
    if not K256_LONG:
        _L().gom_gemm_k256_set_lines(0)                      # (A/B switch only: the round-2 mid-state of the kernel)
    prof = _gemm_profile if _gemm_profile is not None else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().gom_gemm_k256_rp_f32(_p(x), _p(A2), lda, _p(lin.image), _p(R), ldr, rc, int(r_period), 1 if relu else 0, _p(out),
                                    out.stride(0) if out.shape[0] > 1 else N, M, N, lin.K, groups or 1,
                                    _p(range_flag(x.device)), _stream()), "gom_gemm_k256_rp_f32")
    if prof is not None:
        e1.record()
        nbytes = 4.0 * M * lin.K * (2 if A2 is not None else 1) + lin.image.numel() + 4.0 * M * N \
            + 4.0 * (r_period or M) * rc
        prof.append((e0, e1, 2.0 * M * N * lin.K, nbytes, "k256:%dx%dx%d" % (M, N, lin.K), _profile_scope))
    return out


HOIST_MATCH_PROJECTIONS = _switch("HOIST_MATCH_PROJECTIONS")   # native tracker: per-row matcher projections computed once per detection
PROJ_LN = _switch("PROJ_LN")         # f16x3 back-end: out_proj + residual + LayerNorm of every attention block as one launch
POS_PERIODIC = _switch("POS_PERIODIC")   # f16x3 back-end: the encoder's position table read as row m % S (no broadcast copy)


class ProjLN:
    """`LayerNorm(x W^T + b + R)` prepared for gom_proj_ln_f32 (csrc/proj_ln.hip): k-major fragment image of the f16x3 planes of
    W [256, 256] + inverse row scales + bias + the norm's gain / bias.  `pair` / `norm` stay available for the two-launch path."""

    def __init__(self, W, bias, gamma, beta, eps=1e-5):
        assert isinstance(W, SplitWeight) and W.kind == "f16x3"
        nbytes = _L().gom_proj_ln_image_bytes(W.N, W.K)
        if nbytes < 0:
            raise _lib_mod.GomError("projection + LayerNorm kernel does not serve N %d / K %d" % (W.N, W.K))
        _chk_f32(bias, gamma, beta)
        pl = W.planes
        self.image = torch.empty((nbytes,), dtype=torch.uint8, device=pl.device)
        check(_L().gom_proj_ln_image(_p(pl), pl.stride(0), pl.stride(1), W.N, W.K, _p(self.image), nbytes, _stream()),
              "gom_proj_ln_image")
        self.W, self.bias, self.gamma, self.beta, self.eps = W, bias, gamma, beta, eps


def proj_ln_block(pair, norm):
    """(weight, bias) of an attention block's out_proj + (gamma, beta) of the norm behind it -> ProjLN when the back-end and
    shape allow, else None (callers keep the two-launch path)."""
    w, b = pair if not isinstance(pair, K256Linear) else (pair.W, pair.bias)
    if PROJ_LN and GEMM_MODE == "f16x3" and isinstance(w, SplitWeight) and w.kind == "f16x3" and w.N == 256 and w.K == 256:
        return ProjLN(w, b, norm[0], norm[1])
    return None


def proj_ln(x, blk, R, out=None):
    """LayerNorm(x @ W^T + b [+ R]) * gamma + beta in one launch; x, R [M, 256] row-strided (R may be None)."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == 256 and x.dtype == _f32
    assert R is None or (R.shape == x.shape and R.stride(1) == 1 and R.dtype == _f32)
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, 256), dtype=_f32, device=x.device)
    prof = _gemm_profile if (_gemm_profile is not None and M > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().gom_proj_ln_f32(_p(x), x.stride(0) if M > 1 else 256, _p(blk.image), _p(blk.W.inv_scale), _p(blk.bias), _p(R),
                               (R.stride(0) if M > 1 else 256) if R is not None else 0, _p(blk.gamma), _p(blk.beta), blk.eps,
                               _p(out), out.stride(0) if M > 1 else 256, M, _p(range_flag(x.device)), _stream()),
          "gom_proj_ln_f32")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * 256 * 256, (12.0 if R is not None else 8.0) * M * 256 + blk.image.numel(),
                     "projln:%dx256x256" % M, _profile_scope))
    return out


PROPOSAL_DOT = _switch("PROPOSAL_DOT")   # f16x3 back-end: enc_output + norm + class logit of every token as one launch


def proj_ln_dot(x, blk, w, b):
    """[M] = <LayerNorm(x @ W^T + b) * gamma + beta, w> + b in one launch (the normalised rows are not stored); w [256] fp32
    device tensor, b a Python float."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == 256 and x.dtype == _f32
    _chk_f32(w)
    assert w.numel() == 256
    M = x.shape[0]
    out = torch.empty((M,), dtype=_f32, device=x.device)
    prof = _gemm_profile if (_gemm_profile is not None and M > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().gom_proj_ln_dot_f32(_p(x), x.stride(0) if M > 1 else 256, _p(blk.image), _p(blk.W.inv_scale), _p(blk.bias),
                                   _p(blk.gamma), _p(blk.beta), blk.eps, _p(w), float(b), _p(out), M, _p(range_flag(x.device)),
                                   _stream()), "gom_proj_ln_dot_f32")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * 256 * 257, 4.0 * M * 257 + blk.image.numel(), "projdot:%dx256x257" % M, _profile_scope))
    return out


BNECK_FUSED = _switch("BNECK_FUSED")   # f16x3 back-end: a bottleneck block's conv3 + residual + ReLU fused with the next block's conv1


BNECK2 = _switch("BNECK2")           # res4's bottleneck tails + next heads (256 -> 1024 -> 256) fused on csrc/bneck2.hip (two workgroups per CU)


class BneckFused:
    """conv3 (1x1, [c4, k1]) + folded BatchNorm + residual + ReLU of one ResNet bottleneck block and conv1 (1x1, [mp, c4]) +
    folded BatchNorm + ReLU of the NEXT block prepared for gom_bneck_f32 (csrc/bneck_fused.hip): the block's output is written once
    and never read back."""

    def __init__(self, w3, scale3, shift3, w1, scale1, shift1):
        assert isinstance(w3, SplitWeight) and isinstance(w1, SplitWeight) and w3.kind == "f16x3" and w1.kind == "f16x3"
        self.k1, self.c4, self.mp = w3.K, w3.N, w1.N
        assert w1.K == self.c4
        _chk_f32(scale3, shift3, scale1, shift1)
        p3, p1 = w3.planes, w1.planes
        self.v2 = BneckFused._wide(self.k1, self.c4, self.mp)
        if self.v2:
            # res4 (256 -> 1024 -> 256) and the res3 -> res4 transition (128 -> 512 -> 256): csrc/bneck2.hip, 16-pixel waves, eight per
            # workgroup sharing one weight ring
            nbytes = _L().gom_bneck2_image_bytes(self.k1, self.c4, self.mp)
            self.image = torch.empty((nbytes,), dtype=torch.uint8, device=p3.device)
            check(_L().gom_bneck2_image(_p(p3), p3.stride(0), p3.stride(1), _p(w3.inv_scale), _p(scale3), _p(shift3), _p(p1), p1.stride(0),
                                        p1.stride(1), self.k1, self.c4, self.mp, _p(self.image), nbytes, _stream()), "gom_bneck2_image")
            self.sc1 = (scale1 * w1.inv_scale).contiguous()
            self.sh1 = shift1.contiguous()
            return
        nbytes = _L().gom_bneck_image_bytes(self.k1, self.c4, self.mp)
        if nbytes < 0:
            raise _lib_mod.GomError("fused bottleneck kernel does not serve %d -> %d -> %d" % (self.k1, self.c4, self.mp))
        self.image = torch.empty((nbytes,), dtype=torch.uint8, device=p3.device)
        check(_L().gom_bneck_image(_p(p3), p3.stride(0), p3.stride(1), _p(w3.inv_scale), _p(scale3), _p(shift3), _p(p1), p1.stride(0),
                                   p1.stride(1), self.k1, self.c4, self.mp, _p(self.image), nbytes, _stream()), "gom_bneck_image")
        self.sc1 = (scale1 * w1.inv_scale).contiguous()          # exact: the row scale is a power of two (the tile kernel's product)
        self.sh1 = shift1.contiguous()

    @staticmethod
    def _wide(k1, c4, mp):
        return bool(BNECK2) and _L().gom_bneck2_image_bytes(k1, c4, mp) > 0

    @staticmethod
    def serves(w3, w1):
        return (BNECK_FUSED and GEMM_MODE == "f16x3" and isinstance(w3, SplitWeight) and isinstance(w1, SplitWeight)
                and w3.kind == "f16x3" and w1.kind == "f16x3" and w1.K == w3.N
                and (BneckFused._wide(w3.K, w3.N, w1.N) or _L().gom_bneck_image_bytes(w3.K, w3.N, w1.N) > 0))


def bneck_fused(a, blk, R):
    """(X, Y1) = (relu(bn3(conv3(a)) + R), relu(bn1'(conv1'(X)))) in one launch; a [B, H, W, k1], R [B, H, W, c4] NHWC."""
    _chk_f32(a, R)
    B, H, W, k1 = a.shape
    assert k1 == blk.k1 and tuple(R.shape) == (B, H, W, blk.c4) and a.is_contiguous() and R.is_contiguous()
    M = B * H * W
    X = torch.empty((B, H, W, blk.c4), dtype=_f32, device=a.device)
    Y1 = torch.empty((B, H, W, blk.mp), dtype=_f32, device=a.device)
    prof = _gemm_profile if (_gemm_profile is not None and M > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = _L().gom_bneck2_f32 if blk.v2 else _L().gom_bneck_f32
    check(fn(_p(a), k1, _p(blk.image), _p(R), blk.c4, _p(blk.sc1), _p(blk.sh1), _p(X), blk.c4, _p(Y1), blk.mp, M,
             blk.k1, blk.c4, blk.mp, _p(range_flag(a.device)), _stream()), "gom_bneck2_f32" if blk.v2 else "gom_bneck_f32")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * blk.c4 * (blk.k1 + blk.mp), 4.0 * M * (blk.k1 + 2 * blk.c4 + blk.mp) + blk.image.numel(),
                     "bneck:%dx%dx%dx%d" % (M, blk.k1, blk.c4, blk.mp), _profile_scope))
    return X, Y1


DEC_ATTN = _switch("DEC_ATTN")       # f16x3 back-end: the decoder's intra / inter self-attention blocks as one launch each
DEC_ATTN_INTRA = _switch("DEC_ATTN_INTRA")
DEC_ATTN_INTER = _switch("DEC_ATTN_INTER")


class DecAttnBlock:
    """One self-attention block of a composite decoder layer -- in_proj (q | k | v), 8 x 32 attention, out_proj, residual,
    LayerNorm -- prepared for gom_dec_attn_f32 (csrc/dec_attn.hip): the fragment-linear image of in_proj_weight [768, 256] and
    out_proj.weight [256, 256] in the kernel's stage order + what its epilogue needs.  `inter`: attention over the queries of a
    (frame, point) (deformable_transformer.py:396-404) instead of over the points of a query (:386-394)."""

    def __init__(self, in_w, in_b, out_w, out_b, gamma, beta, inter, eps=1e-5, raw=None, form=None):
        """raw = (weight [384, 256] as a SplitWeight or fp32 tensor, bias [384]) of the cross attention's sampling_offsets |
        attention_weights layers: an inter block then also serves `dec_attn(..., raw_pos=query_pos)`.  form 2 (default, `DEC_ATTN2`):
        16-token waves, two per SIMD (csrc/dec_attn2.hip); the form-1 image is kept beside it (dec_inter_heads reads its stages)."""
        self.form = form if form is not None else (2 if DEC_ATTN2 else 1)
        assert tuple(in_w.shape) == (768, 256) and tuple(out_w.shape) == (256, 256)
        nbytes = _L().gom_dec_attn_image_bytes(256, 8)
        if nbytes < 0:
            raise _lib_mod.GomError("decoder attention kernel serves d_model 256 / 8 heads only")
        if raw is not None:
            assert inter and tuple(raw[0].shape) == (384, 256)
            nbytes = _L().gom_dec_attn_raw_image_bytes()
        si = in_w if isinstance(in_w, SplitWeight) else split_weight(in_w.contiguous(), kind="f16x3")
        so = out_w if isinstance(out_w, SplitWeight) else split_weight(out_w.contiguous(), kind="f16x3")
        assert si.kind == "f16x3" and so.kind == "f16x3"
        _chk_f32(in_b, out_b, gamma, beta)
        self.image = torch.empty((nbytes,), dtype=torch.uint8, device=si.planes.device)
        pi, po = si.planes, so.planes
        check(_L().gom_dec_attn_image(_p(pi), pi.stride(0), pi.stride(1), _p(si.inv_scale), _p(in_b), _p(po), po.stride(0),
                                      po.stride(1), _p(so.inv_scale), _p(out_b), _p(gamma), _p(beta), 1 if inter else 0,
                                      _p(self.image), nbytes, _stream()), "gom_dec_attn_image")
        self.eps, self.inter, self.has_raw = eps, bool(inter), raw is not None
        if raw is not None:
            sr = raw[0] if isinstance(raw[0], SplitWeight) else split_weight(raw[0].contiguous(), kind="f16x3")
            assert sr.kind == "f16x3"
            _chk_f32(raw[1])
            check(_L().gom_dec_attn_raw_image(_p(sr.planes), sr.planes.stride(0), sr.planes.stride(1), _p(sr.inv_scale), _p(raw[1]),
                                              _p(self.image), nbytes, _stream()), "gom_dec_attn_raw_image")
        if self.form == 2:
            n2 = _L().gom_dec_attn2_raw_image_bytes() if raw is not None else _L().gom_dec_attn2_image_bytes(256, 8)
            self.image2 = torch.empty((n2,), dtype=torch.uint8, device=si.planes.device)
            check(_L().gom_dec_attn2_image(_p(pi), pi.stride(0), pi.stride(1), _p(si.inv_scale), _p(in_b), _p(po), po.stride(0),
                                           po.stride(1), _p(so.inv_scale), _p(out_b), _p(gamma), _p(beta), 1 if inter else 0,
                                           _p(self.image2), n2, _stream()), "gom_dec_attn2_image")
            if raw is not None:
                check(_L().gom_dec_attn2_raw_image(_p(sr.planes), sr.planes.stride(0), sr.planes.stride(1), _p(sr.inv_scale), _p(raw[1]),
                                                   _p(self.image2), n2, _stream()), "gom_dec_attn2_raw_image")


DEC_ATTN_RAW = _switch("DEC_ATTN_RAW")      # the inter block's launch also makes the cross attention's offsets | logits
DEC_ATTN2 = _switch("DEC_ATTN2")            # the two blocks on 16-token waves, two per SIMD (csrc/dec_attn2.hip, round 6)


def dec_attn_block(in_w, in_b, out_pair, norm, inter, raw=None):
    """DecAttnBlock when the back-end allows, else None (callers keep the five-launch path)."""
    w, b = out_pair if not isinstance(out_pair, K256Linear) else (out_pair.W, out_pair.bias)
    on = DEC_ATTN and (DEC_ATTN_INTER if inter else DEC_ATTN_INTRA)
    if on and GEMM_MODE == "f16x3" and isinstance(in_w, SplitWeight) and in_w.kind == "f16x3" and tuple(in_w.shape) == (768, 256) \
            and isinstance(w, SplitWeight) and w.kind == "f16x3" and tuple(w.shape) == (256, 256):
        if raw is not None:
            rw, rb = raw if not isinstance(raw, K256Linear) else (raw.W, raw.bias)
            raw = (rw, rb) if (DEC_ATTN_RAW and inter and isinstance(rw, SplitWeight) and rw.kind == "f16x3" and tuple(rw.shape) == (384, 256)) else None
        return DecAttnBlock(in_w, in_b, w, b, norm[0], norm[1], inter, raw=raw)
    return None


def dec_attn(x, blk, groups, group_tokens, inner=1, pos=None, out=None, raw_pos=None):
    """LayerNorm(x + out_proj(MHA(...))) of one decoder self-attention block in one launch.  x [rows, 256]; intra (blk.inter
    False): `groups` runs of `group_tokens` <= 32 consecutive rows, q = k = x + pos, v = x; inter: token t of group g is row
    ((g // inner) * group_tokens + t) * inner + g % inner, q = k = v = x.  raw_pos (inter blocks built with `raw`): returns
    (out, raw [rows, 384]) with raw = (out + raw_pos) Wraw^T + braw, the cross attention's sampling offsets | attention logits."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == 256 and x.dtype == _f32
    assert (pos is None) == blk.inter
    if raw_pos is not None:
        assert blk.inter and blk.has_raw and raw_pos.shape == x.shape and raw_pos.stride(1) == 1 and raw_pos.dtype == _f32
        rows = groups * group_tokens
        assert x.shape[0] == rows
        if out is None:
            out = torch.empty((rows, 256), dtype=_f32, device=x.device)
        raw = torch.empty((rows, 384), dtype=_f32, device=x.device)
        prof = _gemm_profile if (_gemm_profile is not None and rows > 0) else None
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        fn, img = (_L().gom_dec_attn2_raw_f32, blk.image2) if blk.form == 2 else (_L().gom_dec_attn_raw_f32, blk.image)
        check(fn(_p(x), x.stride(0) if rows > 1 else 256, _p(img), blk.eps, _p(out),
                 out.stride(0) if rows > 1 else 256, _p(raw_pos), raw_pos.stride(0) if rows > 1 else 256,
                 _p(raw), 384, groups, group_tokens, inner, _p(range_flag(x.device)), _stream()),
              "gom_dec_attn_raw_f32")
        if prof is not None:
            e1.record()
            flops = 2.0 * rows * 256 * (1024 + 384) + 4.0 * rows * group_tokens * 256
            prof.append((e0, e1, flops, 4.0 * rows * (256 * 3 + 384) + blk.image.numel(), "decattn:inter+raw:%dx%d" % (groups, group_tokens),
                         _profile_scope))
        return out, raw
    if pos is not None:
        assert pos.shape == x.shape and pos.stride(1) == 1 and pos.dtype == _f32
    rows = groups * group_tokens
    assert x.shape[0] >= rows
    if out is None:
        out = torch.empty((x.shape[0], 256), dtype=_f32, device=x.device)
    prof = _gemm_profile if (_gemm_profile is not None and rows > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn, img = (_L().gom_dec_attn2_f32, blk.image2) if blk.form == 2 else (_L().gom_dec_attn_f32, blk.image)
    check(fn(_p(x), x.stride(0) if x.shape[0] > 1 else 256, _p(pos),
             (pos.stride(0) if pos.shape[0] > 1 else 256) if pos is not None else 0, _p(img), blk.eps, _p(out),
             out.stride(0) if out.shape[0] > 1 else 256, groups, group_tokens, inner,
             1 if blk.inter else 0, _p(range_flag(x.device)), _stream()), "gom_dec_attn_f32")
    if prof is not None:
        e1.record()
        # the block's nn.Linear products (in_proj 768 + out_proj 256 columns) + QK^T and PV of every head
        flops = 2.0 * rows * 256 * 1024 + 4.0 * rows * group_tokens * 256
        nbytes = 4.0 * rows * 256 * (3 if pos is not None else 2) + blk.image.numel()
        prof.append((e0, e1, flops, nbytes, "decattn:%s:%dx%d" % ("inter" if blk.inter else "intra", groups, group_tokens), _profile_scope))
    return out


def dec_inter_heads(x, blk, groups, group_tokens, inner, out=None):
    """in_proj + attention core of the inter-instance block for 128 < group_tokens <= 352 (csrc/dec_inter.hip): returns the
    concatenated head outputs [rows, 256] (out_proj + residual + LayerNorm: `proj_ln`).  Row mapping as `dec_attn` (inter)."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == 256 and x.dtype == _f32 and blk.inter
    rows = groups * group_tokens
    assert x.shape[0] >= rows
    if out is None:
        out = torch.empty((x.shape[0], 256), dtype=_f32, device=x.device)
    prof = _gemm_profile if (_gemm_profile is not None and rows > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().gom_dec_inter_heads_f32(_p(x), x.stride(0) if x.shape[0] > 1 else 256, _p(blk.image), _p(out),
                                       out.stride(0) if out.shape[0] > 1 else 256, groups, group_tokens, inner,
                                       _p(range_flag(x.device)), _stream()), "gom_dec_inter_heads_f32")
    if prof is not None:
        e1.record()
        flops = 2.0 * rows * 256 * 768 + 4.0 * rows * group_tokens * 256      # in_proj + QK^T and PV of every head
        prof.append((e0, e1, flops, 4.0 * rows * 256 * 2 + 24 * 36864, "decattn:inter-heads:%dx%d" % (groups, group_tokens), _profile_scope))
    return out


DEC_INTER_MAX_FUSED = 128            # tokens per group the one-launch block (csrc/dec_attn.hip) serves; up to 352: csrc/dec_inter.hip
DEC_INTER_MAX_HEADS = 352
FUSED_FFN = _switch("FUSED_FFN")     # f16x3 back-end: FFN blocks as one fused launch (False: GEMM, GEMM, LayerNorm)


class FusedFFN:
    """Weights of one `linear1 -> ReLU -> linear2 -> + residual -> LayerNorm` block prepared for gom_ffn_fused_ln_f32: the
    fragment-linear image of both weight matrices + what the epilogue needs.  Built once per layer (f16x3 mode only)."""

    def __init__(self, w1, b1, w2, b2, gamma, beta, eps=1e-5):
        F_, D_ = w1.shape
        assert w2.shape == (D_, F_)
        nbytes = _L().gom_ffn_fused_image_bytes(D_, F_)
        if nbytes < 0:
            raise _lib_mod.GomError("fused FFN kernel does not serve d_model %d / d_hidden %d" % (D_, F_))
        s1, s2 = split_weight(w1.contiguous(), kind="f16x3"), split_weight(w2.contiguous(), kind="f16x3")
        self.image = torch.empty((nbytes,), dtype=torch.uint8, device=w1.device)
        p1, p2 = s1.planes, s2.planes
        check(_L().gom_ffn_fused_image(_p(p1), p1.stride(0), p1.stride(1), _p(s1.inv_scale), _p(b1), _p(p2), p2.stride(0),
                                       p2.stride(1), D_, F_, _p(self.image), nbytes, _stream()), "gom_ffn_fused_image")
        self.inv2, self.b2, self.gamma, self.beta, self.eps = s2.inv_scale, b2, gamma, beta, eps
        self.D, self.F = D_, F_


FUSED_MLP2 = _switch("FUSED_MLP2")   # f16x3 back-end: the decoder's two-layer 256 -> 256 -> 256 perceptrons as one launch


class FusedMLP2:
    """linear -> ReLU -> linear [-> ReLU] with 256 inputs and outputs prepared for gom_mlp2_fused_f32 (the fused FFN kernel's
    pipeline without residual / LayerNorm)."""

    def __init__(self, w1, b1, w2, b2, relu_out):
        F_, D_ = w1.shape
        assert tuple(w2.shape) == (D_, F_) and D_ == 256
        nbytes = _L().gom_ffn_fused_image_bytes(D_, F_)
        if nbytes < 0:
            raise _lib_mod.GomError("fused two-layer perceptron does not serve %d -> %d -> %d" % (D_, F_, D_))
        s1, s2 = split_weight(w1.contiguous(), kind="f16x3"), split_weight(w2.contiguous(), kind="f16x3")
        self.image = torch.empty((nbytes,), dtype=torch.uint8, device=w1.device)
        p1, p2 = s1.planes, s2.planes
        check(_L().gom_ffn_fused_image(_p(p1), p1.stride(0), p1.stride(1), _p(s1.inv_scale), _p(b1), _p(p2), p2.stride(0),
                                       p2.stride(1), D_, F_, _p(self.image), nbytes, _stream()), "gom_ffn_fused_image")
        self.inv2, self.b2, self.relu_out, self.D, self.F = s2.inv_scale, b2.contiguous(), bool(relu_out), D_, F_


def mlp2_block(w1, b1, w2, b2, relu_out):
    """FusedMLP2 when the back-end and shapes allow (fp32 weights [256, 256] twice), else None."""
    if FUSED_MLP2 and FUSED_FFN and GEMM_MODE == "f16x3" and tuple(w1.shape) == (256, 256) and tuple(w2.shape) == (256, 256):
        return FusedMLP2(w1, b1, w2, b2, relu_out)
    return None


def mlp2_fused(x, blk):
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == 256 and x.dtype == _f32
    M = x.shape[0]
    out = torch.empty((M, 256), dtype=_f32, device=x.device)
    prof = _gemm_profile if (_gemm_profile is not None and M > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().gom_mlp2_fused_f32(_p(x), x.stride(0) if M > 1 else 256, _p(blk.image), _p(blk.inv2), _p(blk.b2),
                                  1 if blk.relu_out else 0, _p(out), 256, M, blk.D, blk.F, _p(range_flag(x.device)), _stream()),
          "gom_mlp2_fused_f32")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 4.0 * M * blk.D * blk.F, 8.0 * M * blk.D + blk.image.numel(), "ffn-mlp2:%dx%dx%d" % (M, blk.D, blk.F),
                     _profile_scope))
    return out


def ffn_fused_ln(x, ffn, out=None):
    """LayerNorm(x + FFN(x)) in one launch (csrc/ffn_fused.hip); x [M, 256] row-strided."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == ffn.D and x.dtype == _f32
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, ffn.D), dtype=_f32, device=x.device)
    prof = _gemm_profile if (_gemm_profile is not None and M > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_L().gom_ffn_fused_ln_f32(_p(x), x.stride(0) if M > 1 else ffn.D, _p(ffn.image), _p(ffn.inv2), _p(ffn.b2),
                                    _p(ffn.gamma), _p(ffn.beta), ffn.eps, _p(out), out.stride(0) if M > 1 else ffn.D, M,
                                    ffn.D, ffn.F, _p(range_flag(x.device)), _stream()), "gom_ffn_fused_ln_f32")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 4.0 * M * ffn.D * ffn.F, 8.0 * M * ffn.D + ffn.image.numel(), "ffn%dx%dx%d" % (M, ffn.D, ffn.F), _profile_scope))
    return out


DEC_TAIL = _switch("DEC_TAIL")       # f16x3 back-end: FFN + ctrl_point_coord + reference refinement + the next ref_point_head, one launch


class DecTail:
    """The row-local tail of a composite decoder layer prepared for gom_dec_tail_f32 (csrc/dec_tail.hip): the concatenated
    fragment-linear image of the FFN block, ctrl_point_coord's two hidden layers and ref_point_head (the latter two with their
    first weight in accumulator order), plus the vectors the three epilogues need."""

    def __init__(self, ffn_w, coord_w, qpos_w, dim_t, eps=1e-5, proj_w=None, form=None, waves=None):
        """proj_w = (out_proj weight [256, 256], bias, norm gain, norm bias) of the cross-attention block: the launch then starts from
        the sampled rows (`dec_tail(..., residual=tgt)`).  form 2 (default where the shapes allow, `DEC_TAIL2`): the CU-cooperative
        kernel of csrc/dec_tail2.hip (80 rows per workgroup, the waves split the output columns; `waves` = 4, or 8 = two per SIMD:
        `DEC_TAIL2_WAVES`); form 1: csrc/dec_tail.hip."""
        w1, b1, w2, b2, gamma, beta = ffn_w
        if form is None:
            form = 2 if (DEC_TAIL2 and w1.shape[0] % 128 == 0) else 1
        self.form = form
        self.waves = int(waves) if waves is not None else DEC_TAIL2_WAVES
        if form == 2:
            self._init2(ffn_w, coord_w, qpos_w, dim_t, eps, proj_w)
            return
        (c1, cb1), (c2, cb2), (W3, b3) = coord_w
        (q1, qb1), (q2, qb2) = qpos_w
        F_, D_ = w1.shape
        assert D_ == 256 and tuple(w2.shape) == (D_, F_) and tuple(W3.shape) == (2, 256)
        for w in (c1, c2, q1, q2):
            assert tuple(w.shape) == (256, 256)
        L = _L()
        nbytes = L.gom_dec_tail_image_bytes(D_, F_, 1)
        if nbytes < 0:
            raise _lib_mod.GomError("decoder tail kernel does not serve d_model %d / d_hidden %d" % (D_, F_))
        lin_bytes = L.gom_dec_tail_lin_image_bytes() if proj_w is not None else 0
        nbytes += lin_bytes
        self.image = torch.empty((nbytes,), dtype=torch.uint8, device=w1.device)
        off = 0
        self.proj = None
        if proj_w is not None:
            wo, bo, pg, pb = proj_w
            assert tuple(wo.shape) == (256, 256)
            so = split_weight(wo.contiguous(), kind="f16x3")
            check(L.gom_dec_tail_lin_image(_p(so.planes), so.planes.stride(0), so.planes.stride(1), _p(self.image), lin_bytes, _stream()),
                  "gom_dec_tail_lin_image")
            off = lin_bytes
            self.proj = (so.inv_scale, bo.contiguous(), pg.contiguous(), pb.contiguous())
        keep = []
        for (wa, ba, wb), fn, F_blk in (((w1, b1, w2), L.gom_ffn_fused_image_acc_order if proj_w is not None else L.gom_ffn_fused_image, F_),
                                        ((c1, cb1, c2), L.gom_ffn_fused_image_acc_order, 256),
                                        ((q1, qb1, q2), L.gom_ffn_fused_image_acc_order, 256)):
            sa, sb = split_weight(wa.contiguous(), kind="f16x3"), split_weight(wb.contiguous(), kind="f16x3")
            n = L.gom_ffn_fused_image_bytes(D_, F_blk)
            pa, pb = sa.planes, sb.planes
            check(fn(_p(pa), pa.stride(0), pa.stride(1), _p(sa.inv_scale), _p(ba), _p(pb), pb.stride(0), pb.stride(1), D_, F_blk,
                     ctypes.c_void_p(self.image.data_ptr() + off), n, _stream()), "gom_ffn_fused_image*")
            off += n
            keep.append(sb.inv_scale)
        assert off == nbytes
        self.inv2, self.c_inv2, self.q_inv2 = keep
        self.b2, self.gamma, self.beta = b2.contiguous(), gamma.contiguous(), beta.contiguous()
        self.c_b2, self.q_b2 = cb2.contiguous(), qb2.contiguous()
        self.W3, self.b3, self.dim_t, self.eps, self.D, self.F = W3.contiguous(), b3.contiguous(), dim_t, eps, D_, F_


    def _init2(self, ffn_w, coord_w, qpos_w, dim_t, eps, proj_w):
        """One linear weight stream per wave (csrc/dec_tail2.hip): [out_proj] | FFN | ctrl_point_coord | ref_point_head; an image
        WITHOUT ref_point_head's part serves the last layer (want_qpos=False): the streams of the two differ only in length."""
        w1, b1, w2, b2, gamma, beta = ffn_w
        (c1, cb1), (c2, cb2), (W3, b3) = coord_w
        (q1, qb1), (q2, qb2) = qpos_w
        F_, D_ = w1.shape
        assert D_ == 256 and tuple(w2.shape) == (D_, F_) and tuple(W3.shape) == (2, 256) and F_ % 128 == 0
        L = _L()
        nw = self.waves
        self.wave_bytes = {q: L.gom_dec_tail2_wave_bytes(D_, F_, 1 if proj_w is not None else 0, q, nw) for q in (0, 1)}
        if self.wave_bytes[1] < 0:
            raise _lib_mod.GomError("decoder tail kernel (form 2) does not serve d_model %d / d_hidden %d" % (D_, F_))
        wb = self.wave_bytes[1]
        self.image = torch.empty((nw * wb,), dtype=torch.uint8, device=w1.device)
        off = 0
        blk_bytes = (256 // nw) * 1024                            # a wave's share of a 256 -> 256 layer or of a chunk of 128 hidden units
        self.proj = None
        if proj_w is not None:
            wo, bo, pg, pb = proj_w
            assert tuple(wo.shape) == (256, 256)
            so = split_weight(wo.contiguous(), kind="f16x3")
            check(L.gom_dec_tail2_image_lin(_p(so.planes), so.planes.stride(0), so.planes.stride(1), _p(self.image), wb, off, nw, _stream()),
                  "gom_dec_tail2_image_lin")
            off += blk_bytes
            self.proj = (so.inv_scale, bo.contiguous(), pg.contiguous(), pb.contiguous())
        keep = []
        for (wa, wb_), F_blk in (((w1, w2), F_), ((c1, c2), 256), ((q1, q2), 256)):
            sa, sb = split_weight(wa.contiguous(), kind="f16x3"), split_weight(wb_.contiguous(), kind="f16x3")
            check(L.gom_dec_tail2_image_mlp(_p(sa.planes), sa.planes.stride(0), sa.planes.stride(1), _p(sb.planes), sb.planes.stride(0),
                                            sb.planes.stride(1), F_blk, _p(self.image), wb, off, nw, _stream()), "gom_dec_tail2_image_mlp")
            off += (F_blk // 128) * blk_bytes
            keep.append((sa.inv_scale, sb.inv_scale))
        assert off == wb
        # the last layer's image: the same streams without ref_point_head's part, packed at the shorter stride
        wb0 = self.wave_bytes[0]
        self.image_last = self.image.view(nw, wb)[:, :wb0].contiguous().view(-1)
        (self.inv1, self.inv2), (self.c_inv1, self.c_inv2), (self.q_inv1, self.q_inv2) = keep
        self.b1, self.b2, self.gamma, self.beta = b1.contiguous(), b2.contiguous(), gamma.contiguous(), beta.contiguous()
        self.c_b1, self.c_b2, self.q_b1, self.q_b2 = cb1.contiguous(), cb2.contiguous(), qb1.contiguous(), qb2.contiguous()
        self.W3, self.b3, self.dim_t, self.eps, self.D, self.F = W3.contiguous(), b3.contiguous(), dim_t, eps, D_, F_


DEC_TAIL2 = _switch("DEC_TAIL2")            # ... as the CU-cooperative split-N kernel (csrc/dec_tail2.hip, round 6)
DEC_TAIL2_WAVES = 8 if _switch("DEC_TAIL2_WAVES8") else 4   # ... with eight waves per workgroup (two per SIMD) or four
DEC_TAIL_PROJ = _switch("DEC_TAIL_PROJ")    # ... with the cross-attention's out_proj + norm_cross in front of it


def tail_form2_wins(M, cus=256):
    """Which form of the tail launch is faster at M rows.  A form-2 workgroup (80 rows) takes ~108 us, a form-1 workgroup (128 rows)
    ~148 us -- 0.74 against 0.86 rows per us and CU: form 2 wins by OCCUPANCY where whole rounds of one-per-CU workgroups decide
    (8 x 100 x 25 rows: 250 workgroups in one round against 157 in one), form 1 by throughput where they do not (8 x 300 x 25 rows:
    three rounds of 108 us against two of 148)."""
    def t(rows, us):
        rounds = -(-M // rows) / float(cus)
        return (math.ceil(rounds) if rounds <= 4 else rounds) * us
    return t(80, 108.0) <= t(128, 148.0)


def dec_tail_block(ffn_w, coord_w, qpos_w, dim_t, proj_w=None):
    """DecTail when the back-end and shapes allow, else None (callers keep the four-launch path).  Under `DEC_TAIL2` the block is the
    CU-cooperative form with the round-5 form beside it (`.alt`): `dec_tail` takes the faster one for the call's row count
    (`tail_form2_wins`)."""
    ok = DEC_TAIL and FUSED_FFN and FUSED_MLP2 and REF_UPDATE and GEMM_MODE == "f16x3" and ffn_w[0].shape[1] == 256 and \
        ffn_w[0].shape[0] % 32 == 0 and all(tuple(w.shape) == (256, 256) for w in (coord_w[0][0], coord_w[1][0], qpos_w[0][0], qpos_w[1][0]))
    if proj_w is not None and not (DEC_TAIL_PROJ and tuple(proj_w[0].shape) == (256, 256)):
        proj_w = None
    if not ok:
        return None
    blk = DecTail(ffn_w, coord_w, qpos_w, dim_t, proj_w=proj_w)
    blk.alt = DecTail(ffn_w, coord_w, qpos_w, dim_t, proj_w=proj_w, form=1) if blk.form == 2 else None
    return blk


def dec_tail(x, blk, ref, want_qpos=True, residual=None):
    """(tgt [M,256], new_ref [M,2], qpos [M,256] | None) of one launch: LayerNorm(x + FFN(x)), the refined reference points and the
    NEXT layer's query position (deformable_transformer.py:352-369, :484-488, :470-473).  A block built with `proj_w` takes
    x = the rows sampled by the cross attention and residual = tgt in front of that block: out_proj + norm_cross run first."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == 256 and x.dtype == _f32
    assert (residual is not None) == (blk.proj is not None)
    _chk_f32(ref)
    M = x.shape[0]
    assert ref.numel() == 2 * M
    if blk.form == 2 and getattr(blk, "alt", None) is not None and not tail_form2_wins(M):
        blk = blk.alt                                            # many rounds of workgroups: the 128-row form's throughput wins
    out = torch.empty((M, 256), dtype=_f32, device=x.device)
    new_ref = torch.empty_like(ref)
    qpos = torch.empty((M, 256), dtype=_f32, device=x.device) if want_qpos else None
    prof = _gemm_profile if (_gemm_profile is not None and M > 0) else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if blk.form == 2:
        r = residual
        assert r is None or (r.dim() == 2 and r.stride(1) == 1 and r.shape == x.shape and r.dtype == _f32)
        pi, pb, pg, pbe = blk.proj if blk.proj is not None else (None, None, None, None)
        check(_L().gom_dec_tail2_f32(_p(x), x.stride(0) if M > 1 else 256, _p(r), (r.stride(0) if M > 1 else 256) if r is not None else 0,
                                     _p(blk.image if want_qpos else blk.image_last), blk.wave_bytes[1 if want_qpos else 0], blk.F,
                                     _p(pi), _p(pb), _p(pg), _p(pbe), blk.eps, _p(blk.inv1), _p(blk.b1), _p(blk.inv2), _p(blk.b2),
                                     _p(blk.gamma), _p(blk.beta), blk.eps, _p(blk.c_inv1), _p(blk.c_b1), _p(blk.c_inv2), _p(blk.c_b2),
                                     _p(blk.W3), _p(blk.b3), _p(ref), _p(blk.dim_t), _p(blk.q_inv1), _p(blk.q_b1), _p(blk.q_inv2),
                                     _p(blk.q_b2), _p(out), 256, _p(new_ref), _p(qpos), 256, M, blk.waves, _p(range_flag(x.device)), _stream()),
              "gom_dec_tail2_f32")
    elif blk.proj is None:
        check(_L().gom_dec_tail_f32(_p(x), x.stride(0) if M > 1 else 256, _p(blk.image), blk.F, _p(blk.inv2), _p(blk.b2), _p(blk.gamma),
                                    _p(blk.beta), blk.eps, _p(blk.c_inv2), _p(blk.c_b2), _p(blk.W3), _p(blk.b3), _p(ref), _p(blk.dim_t),
                                    _p(blk.q_inv2), _p(blk.q_b2), _p(out), 256, _p(new_ref), _p(qpos), 256, M,
                                    _p(range_flag(x.device)), _stream()), "gom_dec_tail_f32")
    else:
        r = residual
        assert r.dim() == 2 and r.stride(1) == 1 and r.shape == x.shape and r.dtype == _f32
        pi, pb, pg, pbe = blk.proj
        check(_L().gom_dec_tail_proj_f32(_p(x), x.stride(0) if M > 1 else 256, _p(r), r.stride(0) if M > 1 else 256, _p(blk.image), blk.F,
                                         _p(pi), _p(pb), _p(pg), _p(pbe), blk.eps, _p(blk.inv2), _p(blk.b2), _p(blk.gamma), _p(blk.beta),
                                         blk.eps, _p(blk.c_inv2), _p(blk.c_b2), _p(blk.W3), _p(blk.b3), _p(ref), _p(blk.dim_t),
                                         _p(blk.q_inv2), _p(blk.q_b2), _p(out), 256, _p(new_ref), _p(qpos), 256, M,
                                         _p(range_flag(x.device)), _stream()), "gom_dec_tail_proj_f32")
    if prof is not None:
        e1.record()
        flops = 4.0 * M * 256 * blk.F + 4.0 * M * 256 * 256 * (2 if want_qpos else 1) + 4.0 * M * 256 + \
            (2.0 * M * 256 * 256 if blk.proj is not None else 0.0)
        prof.append((e0, e1, flops, 4.0 * M * 256 * (3 if want_qpos else 2) + blk.image.numel(), "dectail:%dx%d%s" % (M, blk.F, "+qpos" if want_qpos else ""),
                     _profile_scope))
    return out, new_ref, qpos


# ------------------------------------------------------------------------------------------ MSDA
def _msda_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, *more):
    """Preconditions of the native op (ms_deform_attn_cuda.cu:28-52) + the dtype code of its dispatch (:64)."""
    ts = (value, sampling_loc, attn_weight) + more
    if value.dtype not in (_f32, torch.float64):
        raise _lib_mod.GomError("ms_deform_attn: float32 or float64 tensors only (got %s)" % value.dtype)
    for t in ts:
        if t.dtype != value.dtype or not t.is_cuda or not t.is_contiguous():
            raise _lib_mod.GomError("expected a contiguous %s CUDA tensor, got %s %s contiguous=%s" % (
                value.dtype, t.dtype, t.device, t.is_contiguous()))
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise _lib_mod.GomError("spatial_shapes / level_start_index must be int64")
    if not (spatial_shapes.is_cuda and level_start_index.is_cuda and spatial_shapes.is_contiguous()
            and level_start_index.is_contiguous()):
        raise _lib_mod.GomError("spatial_shapes / level_start_index must be contiguous CUDA tensors")
    B, S, M, D = value.shape
    _, Lq, M2, Lv, Pn, two = sampling_loc.shape
    if (sampling_loc.shape[0] != B or M2 != M or two != 2 or tuple(attn_weight.shape) != (B, Lq, M, Lv, Pn)
            or spatial_shapes.shape[0] != Lv or level_start_index.shape[0] != Lv):
        raise _lib_mod.GomError("ms_deform_attn: value [B,S,M,D], sampling_loc [B,Lq,M,L,P,2], attn_weight [B,Lq,M,L,P], "
                                "spatial_shapes [L,2], level_start_index [L] disagree")
    return 0 if value.dtype == _f32 else 1, (B, S, M, D, Lv, Lq, Pn)


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """Same signature as the reference's adet._C.ms_deform_attn_forward (im2col_step accepted, unused): any heads / channels /
    levels / points, float32 or float64 (third_party/adet/layers/csrc/DeformAttn/ms_deform_attn_cuda.cu:20-80)."""
    code, dims = _msda_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    B, S, M, D, Lv, Lq, Pn = dims
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    if code == 0:           # the 1:1 entry: the shipped shape on the wave-per-query kernel, anything else on the general one
        check(_L().gom_ms_deform_attn_forward(_p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_loc),
                                              _p(attn_weight), _p(out), B, S, M, D, Lv, Lq, Pn, _stream()),
              "gom_ms_deform_attn_forward")
    else:
        check(_L().gom_ms_deform_attn_forward_any(code, _p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_loc),
                                                  _p(attn_weight), _p(out), B, S, M, D, Lv, Lq, Pn, _stream()),
              "gom_ms_deform_attn_forward_any")
    return out


def ms_deform_attn_forward_any(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """The general kernel whatever the shape (tests compare it with the wave-per-query kernel on the shipped shape)."""
    code, dims = _msda_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    B, S, M, D, Lv, Lq, Pn = dims
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    check(_L().gom_ms_deform_attn_forward_any(code, _p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_loc),
                                              _p(attn_weight), _p(out), B, S, M, D, Lv, Lq, Pn, _stream()),
          "gom_ms_deform_attn_forward_any")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step=64):
    """adet._C.ms_deform_attn_backward (ms_deform_attn_cuda.cu:83-156): (grad_value, grad_sampling_loc, grad_attn_weight)."""
    code, dims = _msda_args(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output)
    B, S, M, D, Lv, Lq, Pn = dims
    if grad_output.numel() != B * Lq * M * D:
        raise _lib_mod.GomError("ms_deform_attn_backward: grad_output must hold [B, Lq, M*D] elements")
    gv, gl, gw = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    check(_L().gom_ms_deform_attn_backward(code, _p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_loc),
                                           _p(attn_weight), _p(grad_output), _p(gv), _p(gl), _p(gw), B, S, M, D, Lv, Lq, Pn,
                                           _stream()),
          "gom_ms_deform_attn_backward")
    return gv, gl, gw


def ms_deform_attn_forward_strided(value2d, batch_stride, shapes, lsi, loc, w, B, Lq):
    """value2d: [B*S, 256] column slice (stride(1)==1) of a wider buffer; reads it in place."""
    assert value2d.stride(1) == 1
    out = torch.empty((B * Lq, 256), dtype=_f32, device=value2d.device)
    check(_L().gom_ms_deform_attn_forward_strided(_p(value2d), batch_stride, value2d.stride(0), _p(shapes), _p(lsi),
                                                  _p(loc), _p(w), _p(out), B, Lq, _stream()),
          "gom_ms_deform_attn_forward_strided")
    return out


MSDA_LANES = _switch("MSDA_LANES")   # fused MSDA: per-sample arithmetic distributed over a head's lanes (same bits, 3x fewer VALU)
_msda_lanes_set = [None]


MSDA_WINDOW = _switch("MSDA_WINDOW")   # encoder calls: level-0 queries served from LDS windows of the value map (same bits)
# ... and the level-1 queries (round 6): 8 x 16 tiles whose level-0 samples are gathered from global memory (their level-0 window
# would be 1 161 lines) and whose level-1 .. 3 samples come from windows: bit-identical, encoder call 830 -> 808 us.  (4 x 8 tiles with
# all four levels from windows were slower: 958 vs 865 us; profiles/r06_msda_level1_windows_and_offset_scale.log)
MSDA_WINDOW_L1 = _switch("MSDA_WINDOW_L1")
_msda_window_set = [None]


# Encoder layers decide ONCE, on their first eager call, whether their level-0 queries are worth the LDS windows: the window kernel
# assumes sampling offsets within its 5-pixel halo; an octet group (8 (query, head) pairs) with one sample outside takes the gather
# path after paying for the windows.  Measured at the bench shape (profiles/r06_msda_level1_windows_and_offset_scale.log): 872 us per
# call with no group falling back against 973 us for the gather kernel alone, 1131 us with 87 % falling back (the synthetic weights'
# offsets x 2) -- break-even near 34 %.  A trained checkpoint's offsets are not known here (none ships); the policy makes the
# choice a measurement instead of an assumption.  GOM_MSDA_WINDOW_POLICY=0: always the windows (round 4-5 behaviour).
MSDA_WINDOW_POLICY = _switch("MSDA_WINDOW_POLICY")
MSDA_WINDOW_MAX_FALLBACK = 0.30


def msda_window_groups(encoder_hw0, B):
    """Octet groups of one encoder call's window launches (8 heads x 8-query groups of every tile, padded tiles included)."""
    h0, w0 = int(encoder_hw0[0]), int(encoder_hw0[1])
    n = -(-h0 // 8) * -(-w0 // 16) * 16
    if MSDA_WINDOW_L1 and len(encoder_hw0) >= 4:
        n += -(-int(encoder_hw0[2]) // 8) * -(-int(encoder_hw0[3]) // 16) * 16
    return B * 8 * n


def msda_fused(raw, ref, value2d, batch_stride, shapes, lsi, B, Lq, valid_ratios=None, encoder_hw0=None, fallback_counter=None):
    """raw [B*Lq, >=384] (offsets | logits, stride(1)==1), ref [B*Lq, 2], value2d [B*S, 256] column slice;
    valid_ratios [4,2] fp32 (Wv/W, Hv/H) for padded batches.  encoder_hw0 = (H0, W0[, H1, W1]): an ENCODER call (query q = token
    q, reference points = the tokens' own positions) -- the level-0 (and level-1) queries then run on the LDS-window kernel
    (csrc/msda.hip).  fallback_counter: a one-word int32 device tensor the window launches of THIS call add their fallback octet
    groups to (zeroed here)."""
    assert raw.stride(1) == 1 and value2d.stride(1) == 1
    _chk_f32(ref, valid_ratios)
    if _msda_lanes_set[0] != MSDA_LANES:                     # library-side switch follows ops.MSDA_LANES
        _L().gom_msda_set_lane_distributed(1 if MSDA_LANES else 0)
        _msda_lanes_set[0] = MSDA_LANES
    out = torch.empty((B * Lq, 256), dtype=_f32, device=raw.device)
    prof = _gemm_profile if _gemm_profile is not None else None
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        S_rows = value2d.shape[0]
        # the value map once + raw offsets | logits + output (fp32); 2 * Lq * 8 heads * 16 samples * 4 corners * 32
        _after = lambda: (e1.record(), prof.append((e0, e1, 2.0 * B * Lq * 8 * 16 * 4 * 32,
                                                    4.0 * (S_rows * 256 + B * Lq * (384 + 256)), "msda:%dx%d" % (B, Lq),
                                                    _profile_scope)))
    else:
        _after = lambda: None
    if valid_ratios is not None:
        check(_L().gom_msda_fused_forward_vr(_p(raw), raw.stride(0), _p(ref), _p(value2d), batch_stride,
                                             value2d.stride(0), _p(shapes), _p(lsi), _p(valid_ratios), _p(out), B, Lq,
                                             _stream()), "gom_msda_fused_forward_vr")
        _after()
        return out
    if encoder_hw0 is not None and MSDA_WINDOW and MSDA_LANES:
        mask = 3 if MSDA_WINDOW_L1 else 1
        if _msda_window_set[0] != mask:
            _L().gom_msda_set_window(mask)
            _msda_window_set[0] = mask
        h1, w1 = (int(encoder_hw0[2]), int(encoder_hw0[3])) if len(encoder_hw0) >= 4 else (0, 0)
        if fallback_counter is not None:
            fallback_counter.zero_()
            _L().gom_msda_window_count_fallbacks(_p(fallback_counter))
        try:
            check(_L().gom_msda_fused_forward_encoder(_p(raw), raw.stride(0), _p(ref), _p(value2d), batch_stride, value2d.stride(0),
                                                      _p(shapes), _p(lsi), _p(out), B, Lq, int(encoder_hw0[0]), int(encoder_hw0[1]),
                                                      h1, w1, _stream()), "gom_msda_fused_forward_encoder")
        finally:
            if fallback_counter is not None:
                _L().gom_msda_window_count_fallbacks(None)
    else:
        check(_L().gom_msda_fused_forward(_p(raw), raw.stride(0), _p(ref), _p(value2d), batch_stride, value2d.stride(0),
                                          _p(shapes), _p(lsi), _p(out), B, Lq, _stream()), "gom_msda_fused_forward")
    _after()
    return out


def msda_prepare(raw, ref, spatial_shapes, ref_levels=1):
    """raw [Q, >=384] (offsets | logits), ref [Q, ref_levels, 2] -> loc [Q,8,4,4,2], w [Q,8,4,4]."""
    _chk_f32(ref)
    Q = raw.shape[0]
    loc = torch.empty((Q, 8, 4, 4, 2), dtype=_f32, device=raw.device)
    w = torch.empty((Q, 8, 4, 4), dtype=_f32, device=raw.device)
    check(_L().gom_msda_prepare(_p(raw), raw.stride(0), _p(ref), ref_levels, _p(spatial_shapes), _p(loc), _p(w), Q,
                                _stream()), "gom_msda_prepare")
    return loc, w


# ------------------------------------------------------------------------------------------ attention
def mha_core(q, k, v, out, outer, inner, heads, head_dim, Lq, Lk, strides):
    arr = (ctypes.c_long * 12)(*[int(s) for s in strides])
    check(_L().gom_mha_core_f32(_p(q), _p(k), _p(v), _p(out), outer, inner, heads, head_dim, Lq, Lk, arr, _stream()),
          "gom_mha_core_f32")
    return out


def mha_core_segments(q, k, v, out, segments, num_segments, heads, head_dim, ld_q, ld_k, ld_v, ld_o, max_Lq, max_Lk):
    """Ragged batch: segments int32 [S,4] (device) = (first query row, Lq, first key row, Lk) over shared matrices."""
    check(_L().gom_mha_core_segments_f32(_p(q), _p(k), _p(v), _p(out), _p(segments), num_segments, heads, head_dim,
                                         ld_q, ld_k, ld_v, ld_o, max_Lq, max_Lk, _stream()), "gom_mha_core_segments_f32")
    return out


# ------------------------------------------------------------------------------------------ glue
def preprocess(images, mean, std):
    _chk_f32(images)
    B, C, H, W = images.shape
    assert C == 3
    out = torch.empty((B, H, W, 4), dtype=_f32, device=images.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(_L().gom_preprocess_nchw_to_nhwc4(_p(images), m, s, _p(out), B, H, W, _stream()), "gom_preprocess")
    return out


_resample_tables = {}


def resample_tables(in_size, out_size, device):
    """Pillow-bilinear coefficient tables of one axis on `device` (cached): (bounds i32 [out,2], kk i32 [out,ks], ks)."""
    key = (in_size, out_size, str(device))
    hit = _resample_tables.get(key)
    if hit is None:
        ks = _L().gom_resample_ksize_bilinear(in_size, out_size)
        if ks <= 0:
            raise ValueError("bad resample sizes %d -> %d" % (in_size, out_size))
        bounds = torch.empty((out_size, 2), dtype=torch.int32)
        kk = torch.empty((out_size, ks), dtype=torch.int32)
        check(_L().gom_resample_coeffs_bilinear(in_size, out_size, _p(bounds), _p(kk), ks), "gom_resample_coeffs")
        hit = (bounds.to(device), kk.to(device), ks)
        _resample_tables[key] = hit
    return hit


def _chk_frames(frames):
    if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous() or frames.dim() != 4 \
            or frames.shape[3] != 3:
        raise ValueError("frames must be a contiguous CUDA uint8 [B,H,W,3] tensor")


def resize_u8(frames, out_h, out_w, flip=False):
    """[B,H,W,3] u8 -> [B,out_h,out_w,3] u8, bit-exact with PIL.Image.resize(BILINEAR)."""
    _chk_frames(frames)
    B, H, W, _ = frames.shape
    xb, xk, xks = resample_tables(W, out_w, frames.device)
    yb, yk, yks = resample_tables(H, out_h, frames.device)
    out = torch.empty((B, out_h, out_w, 3), dtype=torch.uint8, device=frames.device)
    check(_L().gom_resize_bilinear_u8_hwc3(_p(frames), B, H, W, _p(xb), _p(xk), xks, _p(yb), _p(yk), yks, _p(out),
                                           out_h, out_w, int(flip), _stream()), "gom_resize_bilinear_u8_hwc3")
    return out


def ingest(frames, out_h, out_w, mean, std, flip):
    """[B,H,W,3] u8 frames -> normalised [B,out_h,out_w,4] f32 backbone input (resize + flip + (x-mean)/std)."""
    _chk_frames(frames)
    B, H, W, _ = frames.shape
    xb, xk, xks = resample_tables(W, out_w, frames.device)
    yb, yk, yks = resample_tables(H, out_h, frames.device)
    out = torch.empty((B, out_h, out_w, 4), dtype=_f32, device=frames.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(_L().gom_ingest_u8_hwc3_to_nhwc4(_p(frames), B, H, W, _p(xb), _p(xk), xks, _p(yb), _p(yk), yks, m, s,
                                           _p(out), out_h, out_w, int(flip), _stream()), "gom_ingest_u8_hwc3_to_nhwc4")
    return out


STEM_POOL = _switch("STEM_POOL")     # f16x3 back-end: the ResNet stem (conv 7x7 / 2 + BN + ReLU + max-pool 3x3 / 2) as one launch


def stem_conv_pool(x, w, scale=None, shift=None):
    """relu(conv7x7 / stride 2 / pad 3 (x) * scale + shift) followed by max_pool2d(3, 2, 1), one launch (csrc/stem_pool.hip):
    x [B, H, W, 4] fp32, w = prep_conv_weight of the [64, 7, 7, 4] stem weight under the f16x3 back-end.  Equal bit for bit to
    conv2d_nhwc(..., relu=True, stride=2, pad=3) + maxpool3x3s2."""
    assert isinstance(w, SplitWeight) and w.kind == "f16x3" and tuple(w.conv_shape) == (64, 7, 7, 4)
    _chk_f32(x, scale, shift)
    B, H, Wd, Cin = x.shape
    assert Cin == 4 and x.is_contiguous()
    OH, OW = (H + 6 - 7) // 2 + 1, (Wd + 6 - 7) // 2 + 1
    PH, PW = (OH + 2 - 3) // 2 + 1, (OW + 2 - 3) // 2 + 1
    y = torch.empty((B, PH, PW, 64), dtype=_f32, device=x.device)
    pl = w.planes
    check(_L().gom_stem_conv_pool_f32(_p(x), _p(pl), pl.stride(0), pl.stride(1), _p(w.inv_scale), _p(scale), _p(shift), _p(y),
                                      B, H, Wd, _p(range_flag(x.device)), _stream()), "gom_stem_conv_pool_f32")
    return y


def stem_pool_serves(w):
    return bool(STEM_POOL and GEMM_MODE == "f16x3" and isinstance(w, SplitWeight) and w.kind == "f16x3"
                and tuple(getattr(w, "conv_shape", ())) == (64, 7, 7, 4))


def maxpool3x3s2(x):
    _chk_f32(x)
    B, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((B, OH, OW, C), dtype=_f32, device=x.device)
    check(_L().gom_maxpool3x3s2_nhwc_f32(_p(x), _p(y), B, H, W, C, _stream()), "gom_maxpool3x3s2")
    return y


def pos_encoding_into(dim_t, level_embed_row, out_view, H, W, valid_hw=None):
    if valid_hw is not None:
        check(_L().gom_pos_encoding_2d_valid_f32(_p(dim_t), _p(level_embed_row), _p(out_view), H, W, int(valid_hw[0]),
                                                 int(valid_hw[1]), _stream()), "gom_pos_encoding_2d_valid_f32")
        return
    check(_L().gom_pos_encoding_2d_f32(_p(dim_t), _p(level_embed_row), _p(out_view), H, W, _stream()),
          "gom_pos_encoding_2d_f32")


def point_pos_embed(pts, dim_t):
    _chk_f32(pts, dim_t)
    Q = pts.numel() // 2
    out = torch.empty((Q, 256), dtype=_f32, device=pts.device)
    check(_L().gom_point_pos_embed_f32(_p(pts), _p(dim_t), _p(out), Q, _stream()), "gom_point_pos_embed_f32")
    return out


def ref_sigmoid(delta, ref, C):
    _chk_f32(ref)
    Q = ref.numel() // 2
    out = torch.empty((Q, C), dtype=_f32, device=ref.device)
    check(_L().gom_ref_sigmoid_f32(_p(delta), delta.stride(0), _p(ref), _p(out), Q, C, _stream()),
          "gom_ref_sigmoid_f32")
    return out


REF_UPDATE = _switch("REF_UPDATE")   # decoder: last MLP layer (N = 2) + reference refinement + next layer's point embedding, one launch


def ref_update(h, last, ref, dim_t, scale=None, want_pos=True):
    """(new_ref [Q,2], pos [Q,256] | None): sigmoid(h @ W3^T + b3 + inverse_sigmoid(ref)) and the sine embedding of the new
    reference points (times `scale` = the level-0 valid ratios of a padded batch); `last` = the (weight [2,256], bias) pair."""
    W3, b3 = last
    _chk_f32(ref, W3, b3, dim_t)
    assert h.dim() == 2 and h.stride(1) == 1 and h.shape[1] == 256 and h.dtype == _f32 and tuple(W3.shape) == (2, 256)
    Q = ref.numel() // 2
    assert h.shape[0] == Q
    new_ref = torch.empty_like(ref)
    pos = torch.empty((Q, 256), dtype=_f32, device=ref.device) if want_pos else None
    sx, sy = scale if scale is not None else (1.0, 1.0)
    check(_L().gom_ref_update_f32(_p(h), h.stride(0) if Q > 1 else 256, _p(W3), _p(b3), _p(ref), _p(dim_t), float(sx), float(sy),
                                  _p(new_ref), _p(pos), Q, _stream()), "gom_ref_update_f32")
    return new_ref, pos


def proposal_valid(shapes, lsi, S, vshapes=None):
    out = torch.empty((S,), dtype=torch.uint8, device=shapes.device)
    if vshapes is not None:
        check(_L().gom_proposal_valid_masked(_p(shapes), _p(lsi), shapes.shape[0], _p(vshapes), _p(out), S, _stream()),
              "gom_proposal_valid_masked")
        return out
    check(_L().gom_proposal_valid(_p(shapes), _p(lsi), shapes.shape[0], _p(out), S, _stream()), "gom_proposal_valid")
    return out


def zero_padded_tokens_(buf, col0, ncols, shapes, lsi, vshapes, B, S):
    """buf [B*S, ld]: zero columns [col0, col0+ncols) of the tokens outside their level's valid region."""
    _chk_f32(buf)
    check(_L().gom_zero_padded_tokens_f32(_p(buf), buf.stride(0), col0, ncols, _p(shapes), _p(lsi), _p(vshapes),
                                          shapes.shape[0], B, S, _stream()), "gom_zero_padded_tokens_f32")
    return buf


def encoder_reference_points(shapes, lsi, S, vshapes=None):
    out = torch.empty((S, 2), dtype=_f32, device=shapes.device)
    if vshapes is not None:
        check(_L().gom_encoder_reference_points_masked(_p(shapes), _p(lsi), _p(vshapes), shapes.shape[0], _p(out), S,
                                                       _stream()), "gom_encoder_reference_points_masked")
        return out
    check(_L().gom_encoder_reference_points(_p(shapes), _p(lsi), shapes.shape[0], _p(out), S, _stream()),
          "gom_encoder_reference_points")
    return out


def bezier_reference_points(coord_raw, topk_idx, shapes, lsi, bern, B, S, nq, P, compact=False, vshapes=None):
    """coord_raw: [B,S,8] (compact=False) or the selected tokens' rows [B*nq,8] (compact=True)."""
    _chk_f32(coord_raw, bern)
    out = torch.empty((B, nq, P, 2), dtype=_f32, device=coord_raw.device)
    if vshapes is not None:
        check(_L().gom_bezier_reference_points_masked(_p(coord_raw), _p(topk_idx), _p(shapes), _p(lsi), _p(vshapes),
                                                      shapes.shape[0], _p(bern), _p(out), B, S, nq, P,
                                                      1 if compact else 0, _stream()), "gom_bezier_reference_points_masked")
        return out
    check(_L().gom_bezier_reference_points(_p(coord_raw), _p(topk_idx), _p(shapes), _p(lsi), shapes.shape[0],
                                           _p(bern), _p(out), B, S, nq, P, 1 if compact else 0, _stream()),
          "gom_bezier_reference_points")
    return out


def scale_xy_(x, sx, sy):
    """In-place scaling of interleaved (x, y) pairs of a contiguous tensor."""
    _chk_f32(x)
    check(_L().gom_scale_xy_f32(_p(x), x.numel() // 2, float(sx), float(sy), _stream()), "gom_scale_xy_f32")
    return x


def copy_words(src, dst):
    """dst <- src over 32-bit words by a kernel (src: pinned host tensor or device tensor of the same byte size)."""
    nbytes = src.numel() * src.element_size()
    assert nbytes == dst.numel() * dst.element_size() and nbytes % 4 == 0 and src.is_contiguous() and dst.is_contiguous()
    check(_L().gom_copy_words(src.data_ptr(), dst.data_ptr(), nbytes // 4, _stream()), "gom_copy_words")
    return dst


def add(a, b):
    _chk_f32(a, b)
    out = torch.empty_like(a)
    check(_L().gom_add_f32(_p(a), _p(b), _p(out), a.numel(), _stream()), "gom_add_f32")
    return out


def broadcast_rows(src, B):
    _chk_f32(src)
    out = torch.empty((B,) + tuple(src.shape), dtype=_f32, device=src.device)
    check(_L().gom_broadcast_rows_f32(_p(src), _p(out), src.numel(), B, _stream()), "gom_broadcast_rows_f32")
    return out


def topk_tokens(logits, B, S, k, valid=None, invalid_logit=None, with_rows=False):
    """logits: [B*S, ld] (column 0 used) -> int32 [B,k] sorted by value desc."""
    nbytes = _L().gom_topk_workspace_bytes(B, S, k)
    ws = torch.empty((max(nbytes, 8),), dtype=torch.uint8, device=logits.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=logits.device)
    rows = torch.empty((B, k), dtype=torch.int32, device=logits.device) if with_rows else None
    check(_L().gom_topk_tokens(_p(logits), logits.stride(0), _p(valid), _p(invalid_logit), B, S, k, _p(ws), _p(idx),
                               _p(rows), _stream()), "gom_topk_tokens")
    return (idx, rows) if with_rows else idx


def argmax_rows(x):
    rows, V = x.shape
    out = torch.empty((rows,), dtype=torch.int32, device=x.device)
    check(_L().gom_argmax_rows_f32(_p(x), x.stride(0), V, rows, _p(out), _stream()), "gom_argmax_rows_f32")
    return out


def detect_post(cls, recls, ctrl, bd, recs, B, nq, P, img_h, img_w, det_thr, nms_thr, asso_thr):
    dev = cls.device
    # count | keep_idx | scores | boxes share one buffer so the host needs a single D2H copy per step
    small = torch.zeros((B * (1 + 6 * nq),), dtype=torch.int32, device=dev)
    o1, o2, o3 = B, B + B * nq, B + 2 * B * nq
    out = {
        "small": small, "small_layout": (o1, o2, o3),
        "count": small[:o1],
        "keep_idx": small[o1:o2].view(B, nq),
        "scores": small[o2:o3].view(_f32).view(B, nq),
        "boxes": small[o3:].view(_f32).view(B, nq, 4),
        "ctrl": torch.zeros((B, nq, P * 2), dtype=_f32, device=dev),
        "bd": torch.zeros((B, nq, P, 4), dtype=_f32, device=dev),
        "recs": torch.zeros((B, nq, P), dtype=torch.int64, device=dev),
    }
    check(_L().gom_detect_post(_p(cls), cls.stride(0), _p(recls), recls.stride(0) if recls is not None else 0,
                               _p(ctrl), _p(bd), _p(recs), B, nq, P, float(img_h), float(img_w), float(det_thr),
                               float(nms_thr), float(asso_thr), _p(out["count"]), _p(out["keep_idx"]),
                               _p(out["scores"]), _p(out["boxes"]), _p(out["ctrl"]), _p(out["bd"]), _p(out["recs"]),
                               _stream()), "gom_detect_post")
    return out


# ------------------------------------------------------------------------------------------ Swin glue
def gemm_gelu(A, W, bias=None):
    """GELU(A @ W^T + bias): fused into the f16x3 epilogue, a separate in-place pass on the other back-ends."""
    if isinstance(W, SplitWeight) and W.kind == "f16x3":
        return gemm(A, W, bias=bias, relu="gelu")
    return gelu_(gemm(A, W, bias=bias))


def layernorm_any(x, gamma, beta, eps=1e-5):
    _chk_f32(x, gamma, beta)
    D = x.shape[-1]
    out = torch.empty_like(x)
    check(_L().gom_layernorm_any_f32(_p(x), _p(gamma), _p(beta), _p(out), x.numel() // D, D, eps, _stream()),
          "gom_layernorm_any_f32")
    return out


def gelu_(x):
    _chk_f32(x)
    check(_L().gom_gelu_f32(_p(x), x.numel(), _stream()), "gom_gelu_f32")
    return x


def swin_patchify(img):
    """[B,H,W,4] -> ([B*Hp*Wp, 64], Hp, Wp)."""
    _chk_f32(img)
    B, H, W, _ = img.shape
    Hp, Wp = (H + 3) // 4, (W + 3) // 4
    out = torch.empty((B * Hp * Wp, 64), dtype=_f32, device=img.device)
    check(_L().gom_swin_patchify_f32(_p(img), _p(out), B, H, W, _stream()), "gom_swin_patchify_f32")
    return out, Hp, Wp


def swin_window_gather(x, B, H, W, shift):
    _chk_f32(x)
    C = x.shape[-1]
    Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
    out = torch.empty((B * Hp * Wp, C), dtype=_f32, device=x.device)
    check(_L().gom_swin_window_gather_f32(_p(x), _p(out), B, H, W, C, shift, _stream()), "gom_swin_window_gather_f32")
    return out


def swin_window_scatter_add(windows, shortcut, B, H, W, shift):
    _chk_f32(windows, shortcut)
    C = shortcut.shape[-1]
    out = torch.empty_like(shortcut)
    check(_L().gom_swin_window_scatter_add_f32(_p(windows), _p(shortcut), _p(out), B, H, W, C, shift, _stream()),
          "gom_swin_window_scatter_add_f32")
    return out


def swin_patch_merge(x, B, H, W):
    _chk_f32(x)
    C = x.shape[-1]
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    out = torch.empty((B * H2 * W2, 4 * C), dtype=_f32, device=x.device)
    check(_L().gom_swin_patch_merge_f32(_p(x), _p(out), B, H, W, C, _stream()), "gom_swin_patch_merge_f32")
    return out, H2, W2


def swin_window_attention(qkv, bias, mask, windows_per_image, heads):
    _chk_f32(qkv, bias, mask)
    C = qkv.shape[1] // 3
    out = torch.empty((qkv.shape[0], C), dtype=_f32, device=qkv.device)
    check(_L().gom_swin_window_attention_f32(_p(qkv), _p(out), _p(bias), _p(mask), qkv.shape[0] // 49, windows_per_image,
                                             heads, C, _stream()), "gom_swin_window_attention_f32")
    return out


# ------------------------------------------------------------------------------------------ ViTAEv2 glue
def im2col(x, KH, KW, stride, pad, dilation, ldo):
    """[B,H,W,C] -> ([B*OH*OW, ldo], OH, OW): columns (kh, kw, c), zeros beyond KH*KW*C."""
    _chk_f32(x)
    B, H, W, C = x.shape
    OH = (H + 2 * pad - dilation * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dilation * (KW - 1) - 1) // stride + 1
    out = torch.empty((B * OH * OW, ldo), dtype=_f32, device=x.device)
    check(_L().gom_im2col_nhwc_f32(_p(x), _p(out), B, H, W, C, KH, KW, stride, pad, dilation, ldo, _stream()),
          "gom_im2col_nhwc_f32")
    return out, OH, OW


def grouped_conv3x3(x, w, scale, shift, groups, stride=1, silu=False, R=None):
    """x [B,H,W,Cin], w [3,3,Cout,Cin/groups] (tap-major: a wave's lanes read contiguous weights) -> [B,OH,OW,Cout] =
    act(conv*scale + shift) + R (padding 1)."""
    _chk_f32(x, w, scale, shift, R)
    B, H, W, Cin = x.shape
    Cout = w.shape[2]
    assert w.shape == (3, 3, Cout, Cin // groups)
    y = torch.empty((B, (H - 1) // stride + 1, (W - 1) // stride + 1, Cout), dtype=_f32, device=x.device)
    if R is not None:
        assert R.numel() == y.numel()
    check(_L().gom_grouped_conv3x3_nhwc_f32(_p(x), _p(w), _p(scale), _p(shift), _p(R), 3 if silu else 0, _p(y), B, H, W,
                                            Cin, Cout, groups, stride, _stream()), "gom_grouped_conv3x3_nhwc_f32")
    return y


def silu_(x):
    _chk_f32(x)
    check(_L().gom_silu_f32(_p(x), x.numel(), _stream()), "gom_silu_f32")
    return x


def vitae_window_gather(x, B, H, W):
    """tokens [B*H*W, C] -> window rows [B*nW*49, C] over the centred zero-padded grid."""
    _chk_f32(x)
    C = x.shape[-1]
    Hp, Wp = -(-H // 7) * 7, -(-W // 7) * 7
    out = torch.empty((B * Hp * Wp, C), dtype=_f32, device=x.device)
    check(_L().gom_vitae_window_gather_f32(_p(x), _p(out), B, H, W, C, _stream()), "gom_vitae_window_gather_f32")
    return out


def vitae_window_crop(windows, B, H, W, R1=None, R2=None):
    """window rows -> tokens [B*H*W, C] (+ R1 + R2)."""
    _chk_f32(windows, R1, R2)
    C = windows.shape[-1]
    out = torch.empty((B * H * W, C), dtype=_f32, device=windows.device)
    check(_L().gom_vitae_window_crop_f32(_p(windows), _p(R1), _p(R2), _p(out), B, H, W, C, _stream()),
          "gom_vitae_window_crop_f32")
    return out


def vitae_window_attention(qkv, heads):
    _chk_f32(qkv)
    C = qkv.shape[1] // 3
    out = torch.empty((qkv.shape[0], C), dtype=_f32, device=qkv.device)
    check(_L().gom_vitae_window_attention_f32(_p(qkv), _p(out), qkv.shape[0] // 49, heads, C, _stream()),
          "gom_vitae_window_attention_f32")
    return out


def softmax_rows_scaled_(x, cols, scale):
    """In place over the first `cols` columns of a 2-D row-strided view."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == _f32
    check(_L().gom_softmax_rows_scaled_f32(_p(x), x.shape[0], cols, x.stride(0), float(scale), _stream()),
          "gom_softmax_rows_scaled_f32")
    return x


def relu_backward(dy, y):
    _chk_f32(dy, y)
    dx = torch.empty_like(dy)
    check(_L().gom_relu_backward_f32(_p(dy), _p(y), _p(dx), dy.numel(), _stream()), "gom_relu_backward_f32")
    return dx


def softmax_rows_backward(P, dP, cols, scale):
    """dS = scale * P * (dP - rowsum(dP * P)) over the first `cols` columns of same-shape row-strided matrices."""
    assert P.shape == dP.shape and P.stride() == dP.stride() and P.stride(1) == 1
    dS = torch.zeros_like(P)
    check(_L().gom_softmax_rows_backward_f32(_p(P), _p(dP), _p(dS), P.shape[0], cols, P.stride(0), float(scale), _stream()),
          "gom_softmax_rows_backward_f32")
    return dS


def asso_ce(logits, frame_offsets, gt, want_loss=True, grad_scale=None):
    """Per (row, frame) cross entropy with a zero background logit (lstmatcher.py:436-475) -> loss [rows, T] and / or
    dlogits = grad_scale * (softmax - onehot); gt int32 [rows, T] (index in frame | n_t = background | -1 = not counted)."""
    _chk_f32(logits)
    R, T = gt.shape
    assert gt.dtype == torch.int32 and gt.is_contiguous() and frame_offsets.dtype == torch.int32
    loss = torch.empty((R, T), dtype=_f32, device=logits.device) if want_loss else None
    dl = torch.zeros_like(logits) if grad_scale is not None else None
    check(_L().gom_asso_ce_f32(_p(logits), logits.shape[1] if logits.dim() == 2 else 0, _p(frame_offsets), T, _p(gt), R,
                               _p(loss), _p(grad_scale), _p(dl), _stream()), "gom_asso_ce_f32")
    return loss, dl


def sigmoid_focal(x, target, alpha, gamma, want_loss=True, want_grad=True):
    _chk_f32(x, target)
    loss = torch.empty_like(x) if want_loss else None
    dx = torch.empty_like(x) if want_grad else None
    check(_L().gom_sigmoid_focal_f32(_p(x), _p(target), float(alpha), float(gamma), x.numel(), _p(loss), _p(dx), _stream()),
          "gom_sigmoid_focal_f32")
    return loss, dx


def transpose_into(x, out):
    """out[c, r] = x[r, c] for 2-D row-strided views (out may be wider than x has rows)."""
    assert x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1
    assert out.shape[0] == x.shape[1] and out.shape[1] >= x.shape[0]
    check(_L().gom_transpose_f32(_p(x), _p(out), x.shape[0], x.shape[1], x.stride(0), out.stride(0), _stream()),
          "gom_transpose_f32")
    return out


def flash_attention(qkv, B, N, heads):
    """qkv [B*N, 3C] (q | k | v, heads contiguous inside each) -> [B*N, C]: full attention per (image, head), the score
    matrix never leaves the CU (attn_flash.hip; f16x3 products)."""
    _chk_f32(qkv)
    assert qkv.dim() == 2 and qkv.is_contiguous() and qkv.shape[0] == B * N
    C = qkv.shape[1] // 3
    out = torch.empty((B * N, C), dtype=_f32, device=qkv.device)
    f = qkv.view(-1)
    check(_L().gom_flash_attention_f32(_p(f), _p(f[C:]), _p(f[2 * C:]), _p(out), B, N, heads, C // heads, 3 * C, C,
                                       _p(range_flag(qkv.device)), _stream()), "gom_flash_attention_f32")
    return out


# ------------------------------------------------------------------------------------------ tracker
def gather_rows(src, rows):
    n, D = rows.numel(), src.shape[1]
    out = torch.empty((n, D), dtype=_f32, device=src.device)
    check(_L().gom_gather_rows_f32(_p(src), _p(rows), _p(out), n, D, _stream()), "gom_gather_rows_f32")
    return out


def asso_activate(logits, offsets, T):
    n_k, N = logits.shape
    out = torch.empty_like(logits)
    check(_L().gom_asso_activate_f32(_p(logits), logits.stride(0) if n_k > 1 else N, _p(offsets), T, n_k, _p(out),
                                     N, _stream()), "gom_asso_activate_f32")
    return out


def track_score(act, meta, decay, boxes, img_w, img_h, n_k, Np, M, with_iou, max_center_dist):
    traj = torch.empty((n_k, M), dtype=_f32, device=act.device)
    check(_L().gom_track_score_f32(_p(act), act.shape[1], _p(meta), _p(decay), _p(boxes), float(img_w), float(img_h),
                                   n_k, Np, M, 1 if with_iou else 0, float(max_center_dist), _p(traj), _stream()),
          "gom_track_score_f32")
    return traj


def pack_records(pool, row_base, det, frames, nq, feature_dim, num_points, image_size):
    """One launch: a step's detections (the nq-padded arrays of detect_post + their pool rows) -> [frames, nq+1, D] fp32."""
    D = feature_dim + 5 + 7 * num_points
    out = torch.empty((frames, nq + 1, D), dtype=_f32, device=pool.device)
    recs = det["recs"]
    assert recs.dtype == torch.int64 and recs.is_contiguous() and det["ctrl"].is_contiguous() and det["bd"].is_contiguous()
    check(_L().gom_pack_records_f32(_p(pool), pool.stride(0), int(row_base), _p(det["count"]), _p(det["boxes"]),
                                    _p(det["scores"]), _p(det["ctrl"]), _p(det["bd"]), _p(recs), frames, nq, feature_dim,
                                    num_points, float(image_size[0]), float(image_size[1]), _p(out), _stream()),
          "gom_pack_records_f32")
    return out


BATCHED_SHORT_TERM = True  # False: per-pair kernels (kept for the A/B parity test and for > 320 detections per frame)
SHORT_TERM_MAX_PREV = 320


def short_term_pairs(tgt, memory, pairs, row_pair, boxes, img_w, img_h, with_iou, total_rows, max_prev, s_floats):
    """All pairs' S = max(softmax-with-background(q.k^T), IoU) in one launch; returns the packed fp32 buffer."""
    S = torch.empty((max(s_floats, 1),), dtype=_f32, device=tgt.device)
    check(_L().gom_short_term_pairs_f32(_p(tgt), _p(memory), tgt.shape[1], _p(pairs), _p(row_pair), _p(boxes),
                                        float(img_w), float(img_h), 1 if with_iou else 0, total_rows, max_prev, _p(S),
                                        _stream()), "gom_short_term_pairs_f32")
    return S


NATIVE_TRACKER = True      # the per-frame id recurrence of a track_frames call in native code (tracker_rt.hip); False: Python loop
NATIVE_MATCHER = True      # False: compose the match from per-kernel calls in Python (kept for the A/B parity test)


class MatcherLayer(ctypes.Structure):
    """Mirror of `gom_matcher_layer` (include/gomatching_hip.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("in_w", "in_b", "out_w", "out_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b")]


def gemm_small_rows(A, W, bias, out):
    """out = A @ W^T + bias by the tracker's small-GEMM kernel WHATEVER the number of rows (in row chunks): every output has
    the bits that kernel gives it inside a match, which is what lets per-row projections be hoisted out of the match chain."""
    assert A.dim() == 2 and A.stride(1) == 1 and W.dim() == 2 and W.stride(1) == 1 and out.stride(1) == 1
    M, K = A.shape
    N = W.shape[0]
    assert out.shape == (M, N) and K % 4 == 0
    for t_ in (A, W, bias, out):                              # row-strided views are fine (lda / ldw / ldc below)
        assert t_ is None or (t_.dtype == _f32 and t_.is_cuda)
    step = max(8, ((1 << 22) // N) // 8 * 8)
    lda, ldw, ldc = (A.stride(0) if M > 1 else K), (W.stride(0) if N > 1 else K), (out.stride(0) if M > 1 else N)
    for a in range(0, M, step):
        m = min(step, M - a)
        check(_L().gom_gemm_small_f32(_p(A[a:]), None, lda, _p(W), ldw, None, _p(bias), None, 0, 0, _p(out[a:]), ldc, m, N, K,
                                      _stream()), "gom_gemm_small_f32")
    return out


def matcher_layers(layers):
    """layers: list of dicts {"in": (w, b), "out": (w, b), ["lin1": (w, b), "lin2": (w, b)]} of fp32 CUDA tensors ->
    ctypes array for gom_match_scores_f32 (the tensors must outlive it: the caller keeps `layers`)."""
    arr = (MatcherLayer * max(len(layers), 1))()
    for i, L in enumerate(layers):
        for key in ("in", "out", "lin1", "lin2"):
            w, b = L.get(key, (None, None))
            if w is not None:
                _chk_f32(w, b)
            setattr(arr[i], key + "_w", w.data_ptr() if w is not None else None)
            setattr(arr[i], key + "_b", b.data_ptr() if b is not None else None)
    return arr


def match_scores(pool, rows, frame_offsets, meta, boxes, decay, N, T, lo, hi, num_tracks, enc, n_enc, dec, n_dec, d,
                 heads, ffn, img_w, img_h, with_iou, max_center_dist):
    """One FFI crossing for the whole device chain of a match (gom_match_scores_f32): returns traj [hi-lo, num_tracks]."""
    n_k = hi - lo
    nws = _L().gom_match_workspace_floats(N, n_k, d, ffn)
    ws = torch.empty((nws,), dtype=_f32, device=pool.device)
    traj = torch.empty((n_k, num_tracks), dtype=_f32, device=pool.device)
    check(_L().gom_match_scores_f32(_p(pool), pool.stride(0), _p(rows), _p(frame_offsets), _p(meta), _p(boxes),
                                    _p(decay), N, T, lo, hi, num_tracks, enc, n_enc, dec, n_dec, d, heads, ffn,
                                    float(img_w), float(img_h), 1 if with_iou else 0, float(max_center_dist), _p(ws),
                                    nws, _p(traj), _stream()), "gom_match_scores_f32")
    return traj


def match_scores_proj(pool, proj, rows, frame_offsets, meta, boxes, decay, N, T, lo, hi, num_tracks, enc, n_enc, dec, n_dec, d,
                      heads, ffn, img_w, img_h, with_iou, max_center_dist):
    """The match with hoisted projections (proj [pool rows, >= 4 d]: encoder-layer-0 in-projection | decoder-layer-0 query
    projection of every pool row): the chain of 13 launches (gom_match_scores_proj_f32).  Returns traj [hi - lo, num_tracks]."""
    n_k = hi - lo
    nws = _L().gom_match_workspace_floats(N, n_k, d, ffn)
    ws = torch.empty((nws,), dtype=_f32, device=pool.device)
    traj = torch.empty((n_k, num_tracks), dtype=_f32, device=pool.device)
    check(_L().gom_match_scores_proj_f32(_p(pool), pool.stride(0), _p(proj), proj.stride(0), _p(rows), _p(frame_offsets),
                                         _p(meta), _p(boxes), _p(decay), N, T, lo, hi, num_tracks, enc, n_enc, dec, n_dec, d,
                                         heads, ffn, float(img_w), float(img_h), 1 if with_iou else 0, float(max_center_dist),
                                         _p(ws), nws, _p(traj), _stream()), "gom_match_scores_proj_f32")
    return traj


def linear_sum_assignment(cost):
    """Host LSA with SciPy-compatible tie-breaking; cost: 2-D numpy array."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    n = min(nr, nc)
    ri = (ctypes.c_long * max(n, 1))()
    ci = (ctypes.c_long * max(n, 1))()
    rc = _L().gom_linear_sum_assignment(cost.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), nr, nc, ri, ci)
    if rc < 0:
        raise _lib_mod.GomError("gom_linear_sum_assignment failed (%d): cost matrix is infeasible or invalid" % rc)
    return np.asarray(ri[:rc], dtype=np.int64), np.asarray(ci[:rc], dtype=np.int64)
