"""ViTAEv2-S backbone: the CPU restatement against the fixture produced by the reference's own ViTAEv2
module (oracle/gen_golden_vitae.py), and the key table of the synthetic weights."""
import numpy as np
import pytest
import torch

from helpers import golden
from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import canonical_keys, synth_state_dict
from oracle import vitae_oracle


@pytest.fixture(scope="module")
def vitae_sd():
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_vitaev2_backbone"
    return synth_state_dict(cfg, seed=3)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_matches_reference_module(vitae_sd, tag):
    g = golden("vitae_s.npz")
    with torch.no_grad():
        out = vitae_oracle.vitae_v2_s(torch.from_numpy(g["x_" + tag]), vitae_sd)
    for k in ("stage3", "stage4", "stage5"):
        ref = g["%s_%s" % (k, tag)]
        assert tuple(out[k].shape) == ref.shape
        assert float(np.abs(out[k].numpy() - ref).max()) <= 2e-5


def test_key_table_and_channels(vitae_sd):
    cfg = setup_cfg(builtin="icdar15")
    cfg.MODEL.BACKBONE.NAME = "build_vitaev2_backbone"
    keys = canonical_keys(cfg)
    bb = {k: v for k, v in keys.items() if k.startswith("backbone.")}
    n_params = sum(int(np.prod(s)) for k, s in bb.items() if not k.endswith(("running_mean", "running_var")))
    assert 18e6 < n_params < 20e6, n_params                     # ViTAEv2-S: ~19 M parameters
    assert "backbone.0.backbone.layers.2.RC.attn.attn.qkv.bias" not in bb       # Token_transformer: qkv_bias=False
    assert keys["detection_transformer.input_proj.0.0.weight"] == (256, 128, 1, 1)
    assert keys["detection_transformer.input_proj.3.0.weight"] == (256, 512, 3, 3)
    assert all(float(v.min()) > 0 for k, v in vitae_sd.items() if k.endswith("running_var"))


def test_rejects_sizes_the_reference_cannot_run(vitae_sd):
    with pytest.raises(ValueError, match="multiples of 32"):
        vitae_oracle.vitae_v2_s(torch.zeros(1, 3, 90, 130), vitae_sd)
