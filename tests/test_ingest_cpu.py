"""Frame ingest, CPU side: the numpy restatement of Pillow's fixed-point bilinear resampler is
pinned against the installed Pillow, and the C-ABI's [host] coefficient helper against the restatement."""
import ctypes

import numpy as np
import pytest
import torch
from PIL import Image

from gomatching_amd import lib
from gomatching_amd.predictor import resized_shape
from oracle import resample_oracle as RO

CASES = [(72, 128, 100, 178), (90, 160, 50, 89), (64, 64, 64, 100), (37, 53, 111, 20), (200, 300, 67, 100),
         (48, 64, 48, 64), (30, 40, 31, 41), (5, 7, 50, 3), (1, 1, 4, 4), (300, 17, 2, 90)]


@pytest.mark.parametrize("h,w,oh,ow", CASES)
def test_oracle_matches_pillow(h, w, oh, ow):
    g = np.random.default_rng(h * 1000 + w)
    for img in (g.integers(0, 256, size=(h, w, 3), dtype=np.uint8),
                np.full((h, w, 3), 255, np.uint8), (g.random((h, w, 3)) < 0.5).astype(np.uint8) * 255):
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(RO.resize_bilinear_u8(img, oh, ow), ref)


def test_oracle_matches_pillow_video_sizes():
    """The shortest-edge targets of the shipped configs on a 720p source (row block only, to stay fast)."""
    g = np.random.default_rng(0)
    img = g.integers(0, 256, size=(720, 1280, 3), dtype=np.uint8)
    for mn, mx in ((1000, 2000), (1280, 2400), (800, 1333)):
        oh, ow = resized_shape(720, 1280, mn, mx)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        xb, xk = RO.bilinear_coeffs(1280, ow)
        yb, yk = RO.bilinear_coeffs(720, oh)
        rows = slice(0, 24)
        need = int(yb[rows.stop - 1, 0] + yb[rows.stop - 1, 1])
        tmp = RO._pass(img[:need], xb, xk, 1)
        assert np.array_equal(RO._pass(tmp, yb[rows], yk[rows], 0), ref[rows])


@pytest.mark.parametrize("n_in,n_out", [(1280, 1778), (720, 1000), (1080, 1000), (1920, 1778), (7, 50), (50, 7),
                                        (64, 64), (1, 5), (2276, 1280), (3, 1000)])
def test_host_coefficients_match_restatement(n_in, n_out):
    L = lib.load()
    ks = L.gom_resample_ksize_bilinear(n_in, n_out)
    bounds, kk = RO.bilinear_coeffs(n_in, n_out)
    assert ks == kk.shape[1]
    b = torch.empty((n_out, 2), dtype=torch.int32)
    k = torch.empty((n_out, ks), dtype=torch.int32)
    assert L.gom_resample_coeffs_bilinear(n_in, n_out, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(k.data_ptr()),
                                          ks) == 0
    assert np.array_equal(b.numpy(), bounds) and np.array_equal(k.numpy(), kk)
    assert np.all(k.numpy().sum(1) > 0)


def test_host_coefficients_reject_bad_arguments():
    L = lib.load()
    assert L.gom_resample_ksize_bilinear(0, 5) == -1
    b = torch.empty((4, 2), dtype=torch.int32)
    k = torch.empty((4, 3), dtype=torch.int32)
    assert L.gom_resample_coeffs_bilinear(8, 4, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(k.data_ptr()), 3) == 1
