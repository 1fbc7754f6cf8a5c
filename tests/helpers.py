"""Shared builders for the parity tests (configs/weights identical to oracle/gen_golden.py)."""
import os

import numpy as np
import torch

from gomatching_amd.config import setup_cfg
from gomatching_amd.weights import synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MINI_NQ = 12


def mini_cfg(builtin="icdar15", nq=MINI_NQ, voc=None, device="cpu"):
    cfg = setup_cfg(builtin=builtin)
    cfg.MODEL.DEVICE = device
    cfg.MODEL.TRANSFORMER.NUM_QUERIES = nq
    if voc is not None:
        cfg.MODEL.TRANSFORMER.VOC_SIZE = voc
    return cfg


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def e2e_state_dict(cfg, gold):
    bias = {"detection_transformer.ctrl_point_class.0.bias": float(gold["cls_bias"][0])}
    if cfg.MODEL.ROI_HEADS.WITH_RESR:
        bias["roi_heads.rescoring_head.bias"] = float(gold["cls_bias"][1])
    return synth_state_dict(cfg, seed=7, cls_bias=bias)


def t(x):
    return torch.from_numpy(np.asarray(x))
