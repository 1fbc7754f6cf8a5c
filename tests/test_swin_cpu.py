"""Swin-T backbone: the CPU restatement against the fixture produced by the reference's own
SwinTransformer module (oracle/gen_golden_swin.py)."""
import numpy as np
import pytest
import torch

from helpers import golden
from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import synth_state_dict
from oracle import swin_oracle


@pytest.fixture(scope="module")
def swin_sd():
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
    return synth_state_dict(cfg, seed=3)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_matches_reference_module(swin_sd, tag):
    g = golden("swin_tiny.npz")
    with torch.no_grad():
        out = swin_oracle.swin_tiny(torch.from_numpy(g["x_" + tag]), swin_sd)
    for k in ("stage3", "stage4", "stage5"):
        ref = g["%s_%s" % (k, tag)]
        assert tuple(out[k].shape) == ref.shape
        assert float(np.abs(out[k].numpy() - ref).max()) <= 2e-5


def test_shift_mask_and_relative_index_shapes():
    m = swin_oracle.shift_mask(12, 17)
    assert m.shape == (2 * 3, 49, 49) and set(np.unique(m.numpy()).tolist()) <= {0.0, -100.0}
    idx = swin_oracle.relative_position_index()
    assert idx.shape == (49, 49) and int(idx.min()) == 0 and int(idx.max()) == 168


def test_oracle_matches_reference_module_swin_small():
    """Swin-S (depths 2, 2, 18, 2; swin_transformer.py:709-721) against the reference module's outputs."""
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_swin_backbone"
    cfg.MODEL.SWIN.TYPE = "small"
    sd = synth_state_dict(cfg, seed=4)
    assert "backbone.0.backbone.layers.2.blocks.17.norm1.weight" in sd
    g = golden("swin_small.npz")
    with torch.no_grad():
        out = swin_oracle.swin_tiny(torch.from_numpy(g["x_s"]), sd)
    for k in ("stage3", "stage4", "stage5"):
        assert float(np.abs(out[k].numpy() - g[k + "_s"]).max()) <= 2e-5
