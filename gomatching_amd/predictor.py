"""Harness-side counterpart of `GoMBatchPredictor.__call__`
(/root/reference/gomatching/text_track_visualizer.py:295-335): colour-order flip, shortest-edge resize,
float32 CHW conversion, the timing window, short-track removal and rescaling on the last batch; plus the
CTC-style transcription decode (`_ctc_decode_recognition`, :167-182) and boundary->polygon conversion
(`pre_vis_process`, :76-91) that eval.py applies to the returned Instances.

The resize follows Detectron2 `ResizeShortestEdge` semantics (mirrored in the reference at
gom_lstmatcher.py:82-96): scale the short side to MIN_SIZE_TEST, cap the long side at MAX_SIZE_TEST,
round with int(x + 0.5), PIL bilinear on uint8.  It runs on the host, outside the timing window, exactly
as in the reference.
"""
import pickle
import time

import numpy as np
import torch

CTLABELS_37 = list("abcdefghijklmnopqrstuvwxyz0123456789")
CTLABELS_96 = [chr(c) for c in range(32, 127)]


def resized_shape(h, w, min_size, max_size):
    scale = min_size * 1.0 / min(h, w)
    if h < w:
        newh, neww = min_size, scale * w
    else:
        newh, neww = scale * h, min_size
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def resize_shortest_edge(img, min_size, max_size):
    """img: HxWx3 uint8 -> resized uint8 (PIL bilinear, as Detectron2's ResizeTransform.apply_image)."""
    from PIL import Image
    h, w = img.shape[:2]
    nh, nw = resized_shape(h, w, min_size, max_size)
    if (nh, nw) == (h, w):
        return img
    return np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))


def new_time_cost(sync=False):
    tc = {k: 0.0 for k in ("pre_process", "backbone", "detector", "rescore", "tracker", "short_match",
                           "long_match", "post_process", "total_time")}
    if sync:
        tc["_sync"] = True
    return tc


class GoMBatchPredictor:
    def __init__(self, cfg, model, device_ingest=False):
        self.cfg = cfg
        self.model = model
        self.device_ingest = device_ingest       # True: uint8 frames are resized/normalised on the GPU (prepare_device)
        self.input_format = cfg.INPUT.FORMAT
        self.min_size, self.max_size = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST

    def prepare(self, original_frames):
        """Host-side, untimed part of __call__ (:315-324): returns (inputs, (height, width))."""
        if self.input_format == "RGB":
            original_frames = [x[:, :, ::-1] for x in original_frames]
        height, width = original_frames[0].shape[:2]
        frames = [resize_shortest_edge(np.ascontiguousarray(x), self.min_size, self.max_size)
                  for x in original_frames]
        # CHW float32, contiguous (the reference hands over the transposed VIEW, whose `.to(device)` inside the window first
        # makes it contiguous on the host; doing that here, outside the window, lets the upload be one async copy per frame)
        frames = [torch.as_tensor(np.ascontiguousarray(x.astype("float32").transpose(2, 0, 1))) for x in frames]
        inputs = [{"image": x, "height": height, "width": width, "video_id": 0} for x in frames]
        return inputs, (height, width)

    def prepare_device(self, original_frames):
        """`prepare` without the host resize: frames go to HBM as the uint8 HWC arrays read from disk
        (2.8 MB per 1280x720 frame instead of 21 MB of resized fp32) and the model's `preprocess_image` resizes,
        flips, converts and normalises them in one kernel, bit-exact with the PIL path of `prepare`."""
        height, width = original_frames[0].shape[:2]
        flip = self.input_format == "RGB"
        inputs = []
        for x in original_frames:
            h, w = x.shape[:2]
            t = torch.as_tensor(np.ascontiguousarray(x))
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise ValueError("frames must be HxWx3 uint8 arrays")
            inputs.append({"frame_u8": t, "resize_hw": resized_shape(h, w, self.min_size, self.max_size),
                           "flip_channels": flip, "height": height, "width": width, "video_id": 0})
        return inputs, (height, width)

    @torch.no_grad()
    def __call__(self, original_frames, instances, batch_id, id_count, last_batch, time_cost, return_time=False):
        """One batch of a video: host preparation (untimed), then the timed window."""
        prepare = self.prepare_device if self.device_ingest else self.prepare
        inputs, frame_hw = prepare(original_frames)
        return self.run_prepared(inputs, frame_hw, instances, batch_id, id_count, last_batch, time_cost, return_time)

    @torch.no_grad()
    def run_prepared(self, inputs, hw, instances, batch_id, id_count, last_batch, time_cost, return_time=False):
        """The reference's timed window (text_track_visualizer.py:325-334): tracking of the batch and, on a video's last
        batch, short-track removal + rescaling to the source resolution (booked under `post_process`)."""
        window = _Stopwatch()
        instances, id_count = self.model.batch_inference(inputs, batch_id, id_count, instances, time_cost)
        if last_batch:
            post = _Stopwatch()
            if self.model.min_track_len > 0:
                instances = self.model._remove_short_track(instances)
            instances = self.model.batch_postprocess(instances, [hw] * len(instances))
            time_cost["post_process"] += post.seconds()
        return (instances, id_count, window.seconds()) if return_time else (instances, id_count)


class _Stopwatch:
    def __init__(self):
        self.t0 = time.time()

    def seconds(self):
        return time.time() - self.t0


class TextDecoder:
    """Character tables + the CTC-style collapse of the reference visualizer (`_ctc_decode_recognition`,
    text_track_visualizer.py:40-55,167-182): class ids >= voc_size - 1 are blanks, a character is emitted when it follows a
    blank or differs from its predecessor."""

    def __init__(self, voc_size, custom_dict=""):
        self.voc_size = voc_size
        if voc_size == 96:
            table = CTLABELS_96
        elif voc_size == 37:
            table = CTLABELS_37
        else:
            with open(custom_dict, "rb") as fp:
                table = [chr(c) for c in pickle.load(fp)]          # the custom dictionaries store code points
        if len(table) != int(voc_size - 1):
            raise AssertionError("voc_size %d does not fit a dictionary of %d characters" % (voc_size, len(table)))
        self.labels = table
        self._table = np.asarray(table, dtype=object)

    def decode(self, rec):
        ids = np.asarray(rec, dtype=np.int64).reshape(-1)
        if ids.size == 0:
            return ""
        char = ids < self.voc_size - 1
        prev_ids = np.concatenate(([-1], ids[:-1]))
        prev_char = np.concatenate(([False], char[:-1]))
        emit = char & (~prev_char | (ids != prev_ids))             # run-length collapse, runs broken by blanks
        return "".join(self._table[ids[emit]])


def boundary_to_polygon(bd):
    """`pre_vis_process` polygon construction (:82-85): bd [25,4] (top xy | bottom xy) -> [50,2] closed outline."""
    bd = np.asarray(bd)
    top, bottom = np.hsplit(bd, 2)
    return np.vstack([top, bottom[::-1]])


class ClipPipeline:
    """Depth-1 software pipeline over consecutive steps (clips or batches of a stream): the detector of step i+1
    is queued on the caller's stream BEFORE the tracker of step i runs on the model's tracker stream, so the
    tracker's host bookkeeping, small kernels and syncs hide under the next step's detection.
    `finish(handle)` receives a `GoMatching.detect_launch` handle and returns the step's result."""

    def __init__(self, model, finish):
        self.model, self.finish, self.pending = model, finish, None

    def _run(self, h):
        trk = self.model._tracker_stream()
        with torch.cuda.stream(trk):
            res = self.finish(h)
        trk.synchronize()
        return res

    def push(self, inputs, time_cost):
        """Queue detection of `inputs`; returns the result of the PREVIOUS step (None for the first)."""
        h = self.model.detect_launch(inputs, time_cost)
        res = self._run(self.pending) if self.pending is not None else None
        self.pending = h
        return res

    def flush(self):
        res = self._run(self.pending) if self.pending is not None else None
        self.pending = None
        return res
