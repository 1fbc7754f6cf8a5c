"""Generate tests/golden/results_writer.json by running the REFERENCE's own harness functions (container-only).

TEST INFRASTRUCTURE.  Run in the build container, where /root/reference exists:

    python -m oracle.gen_golden_writers

Imports /root/reference/eval.py and gomatching/text_track_visualizer.py unmodified, behind sys.modules
stand-ins for the third-party packages this image lacks (cv2, matplotlib, shapely, Detectron2's visualizer
classes -- none of which the pinned functions call, except cv2.minAreaRect/boxPoints inside
`getBboxesAndLabels_icd131`, whose result `getid_text` discards).  Pinned here:

  * Generate_Json_annotation (eval.py:68-109)     -> XML + JSON text, byte for byte
  * getid_text (eval.py:182-210)                  -> res_*.txt text, byte for byte
  * TextTrackingVisualizer._ctc_decode_recognition / pre_vis_process (text_track_visualizer.py:76-91,167-182)
                                                  -> decoded strings and polygon outlines for voc 37 / 96

Only inputs and emitted text/arrays are committed; no reference source travels.
"""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim                                      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _install_harness_stubs():
    ref_shim.install()
    import xml.etree.ElementTree as ET
    sys.modules.setdefault("xml.etree.cElementTree", ET)          # alias removed from the stdlib in 3.9

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    def min_area_rect(pts):                                       # result unused by the pinned functions
        return ((0.0, 0.0), (0.0, 0.0), 0.0)

    mod("cv2", minAreaRect=min_area_rect, boxPoints=lambda r: np.zeros((4, 2), np.float32))
    mod("matplotlib", patches=types.SimpleNamespace())
    mod("matplotlib.colors")
    mod("matplotlib.font_manager")
    mod("shapely")
    mod("shapely.geometry", LineString=object)

    class _Base:
        pass

    class ColorMode:
        IMAGE, IMAGE_BW = 0, 2

    mod("detectron2.config", get_cfg=lambda: None)
    mod("detectron2.data.detection_utils", read_image=None)
    mod("detectron2.utils.logger", setup_logger=lambda *a, **k: None)
    mod("detectron2.engine")
    mod("detectron2.engine.defaults", DefaultPredictor=_Base)
    mod("detectron2.utils.video_visualizer", VideoVisualizer=_Base, random_color=lambda **k: (0, 0, 0),
        _create_text_labels=None)
    mod("detectron2.utils.visualizer", ColorMode=ColorMode, Visualizer=_Base, VisImage=_Base)
    mod("adet.config", add_deepsolo_cfg=lambda cfg: None)
    mod("gomatching.config", add_gom_config=lambda cfg: None)


def _load_eval():
    spec = importlib.util.spec_from_file_location("_ref_eval", os.path.join(ref_shim.REF_ROOT, "eval.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def synthetic_annotation(seed=5):
    """Three videos' worth of rows in the layout eval.py:363 builds (the third one exercises empty frames,
    non-ASCII text, XML-special characters and a tied majority vote)."""
    g = np.random.default_rng(seed)
    words = ["exit", "open", "24h", "a&b", "<tag>", 'q"uote', "café", "中文", ""]
    videos = {}
    for v, (frames, tracks) in enumerate([(4, 3), (6, 5), (5, 4)]):
        ann = {}
        for f in range(frames):
            rows = []
            for t in range(tracks):
                if g.random() < 0.25 or (v == 2 and f == 1):
                    continue
                box = [int(x) for x in g.integers(0, 1280, size=8)]
                text = words[int(g.integers(0, len(words)))] if g.random() < 0.6 else words[t % len(words)]
                seg = [[[int(a), int(b)] for a, b in g.integers(0, 1280, size=(6, 2))]]
                row = box + [int(t + 1 + 10 * v), text]
                if not (v == 1 and t == 0):                       # one track without segmentation (len 10 rows)
                    row.append(seg)
                rows.append(row)
            ann[str(f + 1)] = rows
        videos["Video_%d_1_2" % (v + 1)] = ann
    return videos


def main():
    _install_harness_stubs()
    ev = _load_eval()
    ev.tqdm = lambda x, *a, **k: x
    out = {"videos": synthetic_annotation(), "json": {}, "xml": {}, "txt": {}}
    with tempfile.TemporaryDirectory() as d:
        xml_dir, json_dir = os.path.join(d, "preds"), os.path.join(d, "jsons")
        os.makedirs(xml_dir)
        os.makedirs(json_dir)
        for name, ann in out["videos"].items():
            ev.Generate_Json_annotation(ann, os.path.join(json_dir, name + ".json"),
                                        os.path.join(xml_dir, "res_%s.xml" % name))
        ev.getid_text(xml_dir)
        for name in out["videos"]:
            out["json"][name] = open(os.path.join(json_dir, name + ".json"), encoding="utf-8").read()
            out["xml"][name] = open(os.path.join(xml_dir, "res_%s.xml" % name)).read()
            out["txt"][name] = open(os.path.join(xml_dir, "res_%s.txt" % name)).read()

    # ---- CTC decode + polygon construction
    vis_mod = ref_shim.load("gomatching.text_track_visualizer")
    g = np.random.default_rng(11)
    out["decode"] = {}
    for voc in (37, 96):
        cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(
            TRANSFORMER=types.SimpleNamespace(VOC_SIZE=voc, CUSTOM_DICT="")))
        vis = vis_mod.TextTrackingVisualizer(None, cfg)
        n = 16
        recs = g.integers(0, voc, size=(n, 25))
        recs[g.random((n, 25)) < 0.35] = voc - 1                   # blanks
        recs[:, 1::3] = recs[:, 0::3][:, :recs[:, 1::3].shape[1]]  # runs of repeated characters
        recs[0] = voc - 1                                         # all blank
        bd = g.uniform(0, 640, size=(n, 25, 4)).astype(np.float32)
        inst = ref_shim.Instances((480, 640))
        inst.recs = torch.as_tensor(recs)
        inst.bd = torch.as_tensor(bd)
        inst.track_ids = torch.arange(n)
        inst.ctrl_points = torch.zeros(n, 50)
        pred = vis.pre_vis_process(inst)
        out["decode"][str(voc)] = {"recs": recs.tolist(), "bd": bd.tolist(), "texts": list(pred.texts),
                                   "polys": [np.asarray(p, dtype=np.float32).tolist() for p in pred.polys]}
    path = os.path.join(GOLD, "results_writer.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
