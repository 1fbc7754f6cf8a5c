"""Config for the GoMatching inference path: the reference's yaml keys without yacs/Detectron2.

Key names and defaults mirror the reference's CfgNode tables so that the
reference's own `configs/*.yaml` files load unchanged:
  * DeepSolo keys  -> /root/reference/third_party/adet/config/defaults.py:70-105
  * tracker keys   -> /root/reference/gomatching/config.py:3-80
  * Detectron2 keys actually read on the path (MODEL.DEVICE, PIXEL_MEAN/STD,
    RESNETS.*, INPUT.MIN/MAX_SIZE_TEST, INPUT.FORMAT)
Only inference-path keys get defaults; unknown keys from a yaml are kept as-is.
"""
import copy
import os

import yaml


class CfgNode(dict):
    """dict with attribute access (read/write), nested."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return _wrap(copy.deepcopy(_unwrap(self)))


def _wrap(d):
    if isinstance(d, dict):
        return CfgNode({k: _wrap(v) for k, v in d.items()})
    return d


def _unwrap(d):
    if isinstance(d, dict):
        return {k: _unwrap(v) for k, v in d.items()}
    return d


_DEFAULTS = {
    "MODEL": {
        "META_ARCHITECTURE": "GoMatching",
        "DEVICE": "cuda",
        "WEIGHTS": "",
        "PIXEL_MEAN": [123.675, 116.280, 103.530],
        "PIXEL_STD": [58.395, 57.120, 57.375],
        "BACKBONE": {"NAME": "build_resnet_backbone"},
        "SWIN": {"TYPE": "tiny", "DROP_PATH_RATE": 0.2},
        "ViTAEv2": {"TYPE": "vitaev2_s", "DROP_PATH_RATE": 0.2},
        "RESNETS": {"DEPTH": 50, "STRIDE_IN_1X1": False, "OUT_FEATURES": ["res3", "res4", "res5"]},
        "TRANSFORMER": {
            "ENABLED": True, "INFERENCE_TH_TEST": 0.4, "AUX_LOSS": True,
            "ENC_LAYERS": 6, "DEC_LAYERS": 6, "DIM_FEEDFORWARD": 1024, "HIDDEN_DIM": 256,
            "DROPOUT": 0.0, "NHEADS": 8, "NUM_QUERIES": 100, "ENC_N_POINTS": 4, "DEC_N_POINTS": 4,
            "POSITION_EMBEDDING_SCALE": 6.283185307179586, "NUM_FEATURE_LEVELS": 4,
            "VOC_SIZE": 37, "CUSTOM_DICT": "", "NUM_POINTS": 25, "TEMPERATURE": 10000,
            "BOUNDARY_HEAD": True,
            "LOSS": {"FOCAL_ALPHA": 0.25, "FOCAL_GAMMA": 2.0, "POINT_CLASS_WEIGHT": 1.0,
                     "POINT_COORD_WEIGHT": 1.0, "POINT_TEXT_WEIGHT": 0.5, "BOUNDARY_WEIGHT": 0.5},
        },
        "ROI_HEADS": {"NAME": "LSTMatcher", "NUM_CLASSES": 1, "PROPOSAL_APPEND_GT": False,
                      "WITH_RESR": True, "IOU_THRESHOLDS": [0.5], "IOU_LABELS": [0, 1]},
        "ASSO_ON": True,
        "ASSO_HEAD": {
            "FC_DIM": 1024, "NUM_FC": 2, "NUM_ENCODER_LAYERS": 1, "NUM_DECODER_LAYERS": 1,
            "NUM_WEIGHT_LAYERS": 2, "NUM_HEADS": 8, "DROPOUT": 0.1, "NORM": False,
            "ASSO_THRESH": 0.1, "ASSO_WEIGHT": 1.0, "NEG_UNMATCHED": False,
            "NO_DECODER_SELF_ATT": True, "NO_ENCODER_SELF_ATT": False, "WITH_TEMP_EMB": False,
            "NO_POS_EMB": False, "ASSO_THRESH_TEST": -1.0, "CTRS_WEIGHT": 1.0,
            "ASSO_WEIGHT_LOCAL": 1.0,
        },
    },
    "INPUT": {"FORMAT": "BGR", "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333,
              "VIDEO": {"TRAIN_LEN": 8, "TEST_LEN": 16}},
    "VIDEO_INPUT": False,
    "VIDEO_TEST": {
        "OVERLAP_THRESH": 0.1, "NOT_MULT_THRESH": False, "MIN_TRACK_LEN": 5,
        "MAX_CENTER_DIST": -1.0, "DECAY_TIME": -1.0, "WITH_IOU": False, "LOCAL_TRACK": False,
        "LOCAL_IOU_ONLY": False, "LOCAL_NO_IOU": False, "NMS_THRESH": 0.5,
    },
    "DATASETS": {"TRAIN": [], "TEST": []},
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def get_cfg():
    """Defaults of the keys the inference path reads (reference: get_cfg + add_deepsolo_cfg + add_gom_config)."""
    return _wrap(copy.deepcopy(_DEFAULTS))


def merge_from_file(cfg, path):
    with open(path, "r") as f:
        y = yaml.safe_load(f) or {}
    d = _unwrap(cfg)
    _merge(d, y)
    cfg.clear()
    cfg.update(_wrap(d))
    return cfg


def merge_from_list(cfg, opts):
    """`--opts KEY.SUB value ...` pairs, values parsed as yaml scalars."""
    assert len(opts) % 2 == 0
    for k, v in zip(opts[0::2], opts[1::2]):
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v
    return cfg


_HERE = os.path.dirname(os.path.abspath(__file__))
BUILTIN = {
    "icdar15": os.path.join(_HERE, "configs", "gom_icdar15.yaml"),
    "pp_dstext": os.path.join(_HERE, "configs", "gompp_dstext.yaml"),
    "bovtext": os.path.join(_HERE, "configs", "gom_bovtext.yaml"),
    # the reference's other five configs/*.yaml (the same inference-path keys; tests/test_all_configs_gpu.py)
    "dstext": os.path.join(_HERE, "configs", "gom_dstext.yaml"),
    "artvideo": os.path.join(_HERE, "configs", "gom_artvideo.yaml"),
    "pp_artvideo": os.path.join(_HERE, "configs", "gompp_artvideo.yaml"),
    "pp_icdar15": os.path.join(_HERE, "configs", "gompp_icdar15.yaml"),
    "pp_bovtext": os.path.join(_HERE, "configs", "gompp_bovtext.yaml"),
}
REFERENCE_YAML = {"icdar15": "GoMatching_ICDAR15.yaml", "pp_dstext": "GoMatching_PP_DSText.yaml", "bovtext": "GoMatching_BOVText.yaml",
                  "dstext": "GoMatching_DSText.yaml", "artvideo": "GoMatching_ArTVideo.yaml", "pp_artvideo": "GoMatching_PP_ArTVideo.yaml",
                  "pp_icdar15": "GoMatching_PP_ICDAR15.yaml", "pp_bovtext": "GoMatching_PP_BOVText.yaml"}


def setup_cfg(config_file=None, opts=(), builtin=None):
    """Mirror of eval.py:212-222 `setup_cfg`, including the ASSO_THRESH_TEST override at :220."""
    cfg = get_cfg()
    if builtin is not None:
        config_file = BUILTIN[builtin]
    if config_file:
        merge_from_file(cfg, config_file)
    merge_from_list(cfg, list(opts))
    cfg.MODEL.ASSO_HEAD.ASSO_THRESH_TEST = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST
    return cfg
