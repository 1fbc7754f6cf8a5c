"""Frame ingest on the GPU: `gom_resize_bilinear_u8_hwc3` / `gom_ingest_u8_hwc3_to_nhwc4` through the
C ABI against Pillow itself and the numpy restatement -- bit-exact (integer work) -- and the predictor's device-ingest
path against its host (PIL) path."""
import numpy as np
import pytest
import torch
from PIL import Image

from helpers import mini_cfg
from oracle import resample_oracle as RO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [(72, 128, 100, 178), (90, 160, 50, 89), (64, 64, 64, 100), (37, 53, 111, 20), (200, 300, 67, 100),
         (48, 64, 48, 64), (5, 7, 50, 3), (1, 1, 4, 4), (300, 17, 2, 90)]


@pytest.mark.parametrize("h,w,oh,ow", CASES)
def test_resize_u8_matches_pillow(h, w, oh, ow):
    from gomatching_amd import ops
    g = np.random.default_rng(h * 1000 + w)
    frames = g.integers(0, 256, size=(3, h, w, 3), dtype=np.uint8)
    frames[1] = 255
    out = ops.resize_u8(torch.as_tensor(frames).to(DEV), oh, ow).cpu().numpy()
    flipped = ops.resize_u8(torch.as_tensor(frames).to(DEV), oh, ow, flip=True).cpu().numpy()
    for b in range(3):
        ref = np.asarray(Image.fromarray(frames[b]).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(out[b], ref)
        assert np.array_equal(flipped[b], ref[:, :, ::-1])


@pytest.mark.parametrize("src,mn,mx", [((720, 1280), 1000, 2000), ((720, 1280), 1280, 2400), ((1080, 1920), 1000, 2000),
                                       ((2160, 3840), 1000, 2000), ((480, 640), 1000, 1333)])
def test_resize_full_size_frames(src, mn, mx):
    """BASELINE's frame sizes (720p -> 1000x1778 and 1280x2276) plus down-scaling sources, whole frame vs Pillow."""
    from gomatching_amd import ops
    from gomatching_amd.predictor import resized_shape
    g = np.random.default_rng(src[0])
    frame = g.integers(0, 256, size=src + (3,), dtype=np.uint8)
    oh, ow = resized_shape(src[0], src[1], mn, mx)
    out = ops.resize_u8(torch.as_tensor(frame[None]).to(DEV), oh, ow).cpu().numpy()[0]
    ref = np.asarray(Image.fromarray(frame).resize((ow, oh), Image.BILINEAR))
    assert out.shape == ref.shape and np.array_equal(out, ref)


@pytest.mark.parametrize("flip", [False, True])
def test_ingest_matches_restatement_and_reference_form(flip):
    """Fused kernel == restatement, and == the reference's two-stage form (PIL resize on the host, then the model's
    normaliser kernel on the f32 CHW tensor), bit for bit."""
    from gomatching_amd import ops
    g = np.random.default_rng(4)
    frames = g.integers(0, 256, size=(2, 90, 160, 3), dtype=np.uint8)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    oh, ow = 125, 222
    out = ops.ingest(torch.as_tensor(frames).to(DEV), oh, ow, mean, std, flip).cpu().numpy()
    assert np.array_equal(out, RO.ingest(frames, oh, ow, mean, std, flip))
    host = []
    for f in frames:
        f = f[:, :, ::-1] if flip else f
        r = np.asarray(Image.fromarray(np.ascontiguousarray(f)).resize((ow, oh), Image.BILINEAR))
        host.append(torch.as_tensor(r.astype("float32").transpose(2, 0, 1)))
    two_stage = ops.preprocess(torch.stack(host).to(DEV).contiguous(), mean, std).cpu().numpy()
    assert np.array_equal(out, two_stage)


def test_ingest_rejects_bad_frames():
    from gomatching_amd import ops
    with pytest.raises(ValueError):
        ops.resize_u8(torch.zeros(1, 8, 8, 3), 4, 4)                        # not uint8 / not CUDA
    with pytest.raises(ValueError):
        ops.resize_u8(torch.zeros(1, 8, 8, 4, dtype=torch.uint8, device=DEV), 4, 4)


def test_predictor_device_ingest_equals_host_path():
    """GoMBatchPredictor(device_ingest=True) returns exactly what the reference-form host path returns."""
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.predictor import GoMBatchPredictor, new_time_cost
    from gomatching_amd.synth import make_clip
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg("icdar15", device=DEV)
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 128, 256
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.5})
    clip = [np.ascontiguousarray(f[:, :, ::-1]) for f in make_clip(5, 72, 128, clip_id=2)]   # "BGR as read from disk"
    results = []
    for device_ingest in (False, True):
        model = GoMatching(cfg, sd, device=DEV, frames_per_step=4)
        spotter = GoMBatchPredictor(cfg, model, device_ingest=device_ingest)
        res, id_count = spotter(clip, [], 0, 0, True, new_time_cost())
        results.append((res, id_count))
    (a, ida), (b, idb) = results
    assert ida == idb and len(a) == len(b) == 5
    assert sum(len(r["instances"]) for r in a) > 0
    for x, y in zip(a, b):
        x, y = x["instances"], y["instances"]
        assert x.image_size == y.image_size
        assert torch.equal(x.track_ids, y.track_ids) and torch.equal(x.recs, y.recs)
        assert torch.equal(x.scores, y.scores) and torch.equal(x.bd, y.bd)


def test_video_to_result_files(tmp_path):
    """The harness end to end on the GPU: uint8 BGR frames -> device ingest -> batch predictor (100-frame chunking) ->
    result writers; the XML / JSON / TXT files carry exactly the tracks the model returned."""
    import json
    import xml.etree.ElementTree as ET
    from gomatching_amd import results
    from gomatching_amd.modeling import GoMatching
    from gomatching_amd.predictor import GoMBatchPredictor, TextDecoder, new_time_cost
    from gomatching_amd.synth import make_clip
    from gomatching_amd.weights import synth_state_dict
    cfg = mini_cfg("icdar15", device=DEV)
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 128, 256
    sd = synth_state_dict(cfg, seed=7, cls_bias={"detection_transformer.ctrl_point_class.0.bias": 0.5})
    model = GoMatching(cfg, sd, device=DEV, frames_per_step=4)
    spotter = GoMBatchPredictor(cfg, model, device_ingest=True)
    frames = [np.ascontiguousarray(f[:, :, ::-1]) for f in make_clip(7, 72, 128, clip_id=3)]
    tc = new_time_cost()
    preds, seconds = results.spot_video(spotter, frames, tc)
    assert len(preds) == 7 and seconds > 0 and tc["total_time"] == seconds
    dec = TextDecoder(cfg.MODEL.TRANSFORMER.VOC_SIZE)
    ann = results.write_video(preds, "Video_5_1_2", "ICDAR15", str(tmp_path), dec)
    results.write_track_transcriptions(str(tmp_path / "preds"))
    root = ET.parse(str(tmp_path / "preds" / "res_video_5.xml")).getroot()
    js = json.loads((tmp_path / "jsons" / "Video_5_1_2.json").read_text(encoding="utf-8"))
    assert [fr.attrib["ID"] for fr in root] == [str(i + 1) for i in range(7)] == list(js)
    n_rows = 0
    for i, fr in enumerate(root):
        rows = ann[str(i + 1)]
        kept_ids = set(int(t) for t in preds[i]["instances"].track_ids.cpu().tolist())
        assert [int(o.attrib["ID"]) for o in fr] == [r[8] for r in rows] == [o["ID"] for o in js[str(i + 1)]]
        assert set(r[8] for r in rows) <= kept_ids                      # writers may only drop (tiny boxes), never add
        for o, r in zip(fr, rows):
            assert [(int(p.attrib["x"]), int(p.attrib["y"])) for p in o] == list(zip(r[0:8:2], r[1:8:2]))
        n_rows += len(rows)
    assert n_rows > 0
    txt = (tmp_path / "preds" / "res_video_5.txt").read_text().splitlines()
    assert sorted(int(l.split(",")[0].strip('"')) for l in txt) == sorted(set(r[8] for rows in ann.values() for r in rows))
