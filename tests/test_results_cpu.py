"""Result writers / scorer hand-off against the fixture produced by the reference's own
Generate_Json_annotation / getid_text / pre_vis_process (oracle/gen_golden_writers.py)."""
import json
import os

import numpy as np
import pytest
import torch

from gomatching_amd import results as R
from gomatching_amd.predictor import TextDecoder, boundary_to_polygon
from gomatching_amd.structures import Instances

GOLD = os.path.join(os.path.dirname(__file__), "golden", "results_writer.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD, encoding="utf-8") as f:
        return json.load(f)


def test_xml_json_txt_bytes(gold, tmp_path):
    xml_dir, json_dir = tmp_path / "preds", tmp_path / "jsons"
    xml_dir.mkdir()
    json_dir.mkdir()
    for name, ann in gold["videos"].items():
        R.write_video_results(ann, str(json_dir / (name + ".json")), str(xml_dir / ("res_%s.xml" % name)))
    R.write_track_transcriptions(str(xml_dir))
    for name in gold["videos"]:
        assert (json_dir / (name + ".json")).read_text(encoding="utf-8") == gold["json"][name]
        assert (xml_dir / ("res_%s.xml" % name)).read_text() == gold["xml"][name]
        assert (xml_dir / ("res_%s.txt" % name)).read_text() == gold["txt"][name]


@pytest.mark.parametrize("voc", [37, 96])
def test_decode_and_polygons(gold, voc):
    case = gold["decode"][str(voc)]
    dec = TextDecoder(voc)
    recs, bd = np.asarray(case["recs"]), np.asarray(case["bd"], dtype=np.float32)
    assert [dec.decode(r) for r in recs] == case["texts"]
    for b, p in zip(bd, case["polys"]):
        assert np.array_equal(boundary_to_polygon(b), np.asarray(p, dtype=np.float32))


def test_min_area_rect_properties():
    g = np.random.default_rng(3)
    for trial in range(50):
        pts = g.uniform(0, 500, size=(int(g.integers(3, 40)), 2))
        (cx, cy), (w, h), ang = R.min_area_rect(pts)
        box = R.box_points(((cx, cy), (w, h), ang)).astype(np.float64)
        # encloses every point (fp32 corner rounding tolerance)
        a = np.radians(ang)
        u, v = np.array([np.cos(a), np.sin(a)]), np.array([-np.sin(a), np.cos(a)])
        d = pts - np.array([cx, cy])
        assert np.all(np.abs(d @ u) <= w / 2 + 1e-6) and np.all(np.abs(d @ v) <= h / 2 + 1e-6)
        # box corners reproduce centre / extents
        assert np.allclose(box.mean(0), [cx, cy], atol=1e-3)
        assert np.isclose(np.linalg.norm(box[3] - box[0]), w, atol=1e-2)
        assert np.isclose(np.linalg.norm(box[1] - box[0]), h, atol=1e-2)
        # no axis-aligned or 1-degree-swept rectangle is smaller
        best = np.inf
        for t in np.radians(np.arange(0, 180, 1.0)):
            uu, vv = np.array([np.cos(t), np.sin(t)]), np.array([-np.sin(t), np.cos(t)])
            best = min(best, np.ptp(pts @ uu) * np.ptp(pts @ vv))
        assert w * h <= best + 1e-6


def test_min_area_rect_axis_aligned():
    rect = R.min_area_rect([[10, 20], [110, 20], [110, 50], [10, 50]])
    assert np.isclose(rect[1][0] * rect[1][1], 3000.0)
    xs = sorted(int(round(x)) for x in R.box_points(rect)[:, 0])
    ys = sorted(int(round(y)) for y in R.box_points(rect)[:, 1])
    assert xs == [10, 10, 110, 110] and ys == [20, 20, 50, 50]


def test_frame_lines_and_write_clip(tmp_path):
    dec = TextDecoder(37)
    inst = Instances((720, 1280))
    top = np.stack([np.linspace(100, 300, 25), np.full(25, 200.0)], 1)
    bot = np.stack([np.linspace(100, 300, 25), np.full(25, 260.0)], 1)
    tiny = np.concatenate([top * 0 + 5, top * 0 + 7], 1)            # extent < 5 px -> dropped (eval.py:359)
    inst.bd = torch.as_tensor(np.stack([np.concatenate([top, bot], 1), tiny]), dtype=torch.float32)
    recs = np.full((2, 25), 36)
    recs[0, :4] = [4, 23, 8, 19]                                     # "exit"
    inst.recs = torch.as_tensor(recs)
    inst.track_ids = torch.as_tensor([7, 9])
    lines = R.frame_lines(inst, dec)
    assert len(lines) == 1
    row = lines[0]
    assert row[8] == 7 and row[9] == "exit" and len(row[10][0]) == 50
    assert min(row[0:8:2]) in (99, 100) and max(row[0:8:2]) in (299, 300)        # int() truncation (eval.py:356)
    assert min(row[1:8:2]) in (199, 200) and max(row[1:8:2]) in (259, 260)
    empty = Instances((720, 1280))
    empty.bd = torch.zeros(0, 25, 4)
    empty.recs = torch.zeros(0, 25, dtype=torch.long)
    empty.track_ids = torch.zeros(0, dtype=torch.long)
    ann = R.write_clip([{"instances": inst}, {"instances": empty}], "Video_9_1_2", str(tmp_path), dec)
    assert list(ann) == ["1", "2"] and ann["2"] == []
    R.write_track_transcriptions(str(tmp_path / "preds"))
    assert (tmp_path / "preds" / "res_Video_9_1_2.txt").read_text() == '"7","exit"\n'
    js = json.loads((tmp_path / "jsons" / "Video_9_1_2.json").read_text(encoding="utf-8"))
    assert js["1"][0]["ID"] == 7 and js["2"] == []


def test_result_names():
    assert R.result_names("Video_35_2_3", "ICDAR15") == ("video_35", "Video_35_2_3")
    assert R.result_names("Video_35_2_3", "DSText") == ("Video_35_2_3", "Video_35_2_3")


def test_spot_video_chunks():
    calls = []

    def predictor(frames, instances, batch_id, id_count, last_batch, time_cost, return_time=False):
        calls.append((len(frames), batch_id, id_count, last_batch, len(instances)))
        return instances + list(frames), id_count + len(frames), 0.5

    tc = {"total_time": 0.0}
    out, seconds = R.spot_video(predictor, list(range(230)), tc, batch=100)
    assert calls == [(100, 0, 0, False, 0), (100, 1, 100, False, 100), (30, 2, 200, True, 200)]
    assert out == list(range(230)) and seconds == 1.5 and tc["total_time"] == 1.5
