"""CPU restatement of the frame-ingest path.  TEST INFRASTRUCTURE: imported only by tests/.

The resize the reference applies to every frame (text_track_visualizer.py:315-324 -> Detectron2
`ResizeShortestEdge` -> `PIL.Image.resize(size, BILINEAR)` on uint8) lives in a third-party dependency that
is absent from /root/reference: Pillow (unpinned by the reference's README; 12.2.0 in this image),
src/libImaging/Resample.c.  Its published algorithm is restated here in numpy integer arithmetic and PINNED
in tests/test_ingest_cpu.py against the installed Pillow itself (bit-exact over up- and down-scaling cases).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0), box = whole axis.
    Returns (bounds int32 [out,2] = (first tap, tap count), kk int32 [out,ksize])."""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    filterscale = scale = float(in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            wx = 1.0 - t if t < 1.0 else 0.0
            w.append(wx)
            ww += wx
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One 8bpc pass along `axis` (0 = vertical, 1 = horizontal) of an HxWxC uint8 image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.int64)
    for o in range(bounds.shape[0]):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[lo + x] * int(kk[o, x])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """HxWx3 uint8 -> out_h x out_w x 3 uint8, horizontal pass then vertical pass (ImagingResample)."""
    h, w = img.shape[:2]
    xb, xk = bilinear_coeffs(w, out_w)
    yb, yk = bilinear_coeffs(h, out_h)
    tmp = _pass(img, xb, xk, 1) if out_w != w else img
    return _pass(tmp, yb, yk, 0) if out_h != h else tmp


def ingest(frames_bgr, out_h, out_w, mean, std, flip):
    """[B,H,W,3] uint8 -> [B,out_h,out_w,4] float32: resize, optional BGR->RGB, (x-mean)/std in fp32, pad channel."""
    out = np.zeros((len(frames_bgr), out_h, out_w, 4), np.float32)
    mean, std = np.asarray(mean, np.float32), np.asarray(std, np.float32)
    for b, f in enumerate(frames_bgr):
        r = resize_bilinear_u8(f, out_h, out_w)
        if flip:
            r = r[:, :, ::-1]
        out[b, :, :, :3] = (r.astype(np.float32) - mean) / std
    return out
