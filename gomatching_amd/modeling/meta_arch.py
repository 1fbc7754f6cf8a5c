"""`GoMatching` meta-architecture on MI355X: the reference's inference interface
(/root/reference/gomatching/modeling/meta_arch/gom_lstmatcher.py:113-651) over the HIP kernels.

Same public surface as the reference class -- `inference`, `batch_inference`, `run_short_term_match`,
`run_long_term_match`, `_remove_short_track`, `batch_postprocess`, `min_track_len`, `roi_heads`,
`detection_transformer` -- so `GoMBatchPredictor.__call__` (text_track_visualizer.py:295-335) and
eval.py drive it unchanged.  What differs is the schedule:
  * detection + embedding of a whole step of frames runs as ONE batch on the GPU (the reference loops
    frames one at a time, :369-371); frames are independent there (frozen BN, per-sample GN/LN);
  * thresholding, boxes, NMS and the foreground filter stay on the device with nq-padded outputs
    (`gom_detect_post`), so there is one host sync per step, not several per frame;
  * the tracker keeps the reference's strictly sequential control flow and host-side integer id
    bookkeeping + assignment; its float arithmetic runs in HIP kernels.
`forward` (training: the loss dict of the trainable head) goes through gomatching_amd/training.py.
"""
import os
import time

import numpy as np
import torch

from .. import ops
from ..structures import Boxes, Instances
from ..weights import normalize_state_dict
from .backbone import ResNet50
from .swin import SwinTiny
from .vitae import ViTAEv2S
from .deepsolo import DeepSolo
from .roi_heads import build_roi_heads

_f32 = torch.float32


class GoMatching:
    def __init__(self, cfg, state_dict, device=None, frames_per_step=8, use_graphs=True):
        self.cfg = cfg
        self.device = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        if self.device.type != "cuda":
            raise RuntimeError("gomatching_amd runs on an MI355X only (MODEL.DEVICE=%s); there is no CPU path" %
                               cfg.MODEL.DEVICE)
        if cfg.MODEL.BACKBONE.NAME not in ("build_resnet_backbone", "build_swin_backbone", "build_vitaev2_backbone"):
            raise NotImplementedError("backbone %s is not built (R-50, Swin-T and ViTAEv2-S are; SURVEY.md §8-f3)"
                                      % cfg.MODEL.BACKBONE.NAME)
        V = cfg.VIDEO_TEST
        self.test_len = cfg.INPUT.VIDEO.TEST_LEN
        self.overlap_thresh = V.OVERLAP_THRESH
        self.min_track_len = V.MIN_TRACK_LEN
        self.max_center_dist = V.MAX_CENTER_DIST
        self.decay_time = V.DECAY_TIME
        self.asso_thresh = cfg.MODEL.ASSO_HEAD.ASSO_THRESH
        self.with_iou = V.WITH_IOU
        self.not_mult_thresh = V.NOT_MULT_THRESH
        self.nms_thresh = V.NMS_THRESH
        self.with_rescore = cfg.MODEL.ROI_HEADS.WITH_RESR
        self.test_score_threshold = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST
        self.min_size_test = None          # only set for the ViTAE backbone in the reference (:144-146)
        self.max_size_test = None
        if cfg.MODEL.BACKBONE.NAME == "build_vitaev2_backbone":
            self.min_size_test, self.max_size_test = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST
        self.pixel_mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self.pixel_std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        self.frames_per_step = frames_per_step
        self.use_graphs = use_graphs                             # hipGraph replay of the detector (see _detect_graphed)
        self._graphs = {}
        self.max_graphs = 2
        # tracker descriptor uploads (see _h2d): "kernel" = a copy kernel reads the device-mapped pinned staging buffer, so
        # the upload is ordered against the match kernels by plain kernel order; "dma" = async hipMemcpy; "sync" = dma + wait
        self.h2d_mode = "kernel"
        self.pinned_d2h = True             # tracker scores land in pinned memory (see _d2h); False: Tensor.cpu()
        self.training = False

        sd = normalize_state_dict(state_dict)
        self._head_state = {k: v for k, v in sd.items() if k.startswith("roi_heads.")}
        # Precision fallback (the reference runs whatever a checkpoint holds, gom_lstmatcher.py:268-351): the f16x3 kernels raise
        # a device flag when an activation leaves fp16's range; `detect_finish` then re-runs that step on a bf16x6 twin of the
        # detector (fp32's exponent range; weights split lazily, once, from the retained state dict), warns and continues.
        self.gemm_mode = ops.GEMM_MODE
        self.precision_fallback = True
        self.fallback_steps = 0                                  # steps re-run on the bf16x6 twin so far
        self._sd = {k: v for k, v in sd.items() if not k.startswith("roi_heads.")}
        self._fallback_det = None
        self.backbone, self.detection_transformer = self._build_detector(self._sd)
        self.feature_names = self.backbone.out_features
        self.roi_heads = build_roi_heads(cfg, sd, self.device)
        self._pool = None
        self._pool_used = 0

    def _build_detector(self, sd):
        """(backbone, DeepSolo) under the current contraction back-end (ops.GEMM_MODE)."""
        cfg = self.cfg
        if cfg.MODEL.BACKBONE.NAME == "build_swin_backbone":
            if cfg.MODEL.SWIN.TYPE not in ("tiny", "small"):
                raise NotImplementedError("Swin-T and Swin-S are built (the types detection_transformer_wobackbone.py:61-64 "
                                          "admits)")
            backbone = SwinTiny(sd, self.device, swin_type=cfg.MODEL.SWIN.TYPE)
        elif cfg.MODEL.BACKBONE.NAME == "build_vitaev2_backbone":
            if cfg.MODEL.ViTAEv2.TYPE != "vitaev2_s":
                raise NotImplementedError("only vitaev2_s exists (detection_transformer_wobackbone.py:64-68)")
            backbone = ViTAEv2S(sd, self.device)
        else:
            backbone = ResNet50(sd, self.device)
        return backbone, DeepSolo(cfg, sd, self.device)

    def add_class_bias(self, shift):
        """Shift the detector's point-class bias (bench / test calibration of random-init weights) on the live detector, in the
        retained state dict and on the fallback twin, so that all of them keep holding the same weights."""
        self.detection_transformer.ctrl_class[1].add_(shift)
        for k in [k for k in self._sd if k.startswith("detection_transformer.ctrl_point_class.") and k.endswith(".bias")]:
            self._sd[k] = self._sd[k] + shift
        if self._fallback_det is not None:
            self._fallback_det[1].ctrl_class[1].add_(shift)

    @classmethod
    def from_config(cls, cfg, state_dict, **kw):
        return cls(cfg, state_dict, **kw)

    def eval(self):
        return self

    def __call__(self, batched_inputs):
        return self.forward(batched_inputs)

    def forward(self, batched_inputs):
        """The reference's training entry (gom_lstmatcher.py:213-266): {'loss_long_asso', 'loss_short_asso'[, 'loss_res']} with
        autograd history onto `trainable_parameters()` (only `roi_heads` trains, freeze_layers.py:20-37).  The frozen detector
        runs on the inference kernels, the head's forward / backward on the HIP training path (gomatching_amd/training.py).
        Under Detectron2 use the nn.Module wrapper `compat.d2_register.GoMatchingMI355X`, whose parameters are the live ones."""
        from .. import training
        return training.forward_losses(self, batched_inputs)

    def trainable_parameters(self):
        """{state-dict key: leaf tensor with requires_grad} of the trainable head, created from the loaded weights on first use.
        Inference keeps using the weights the model was built with; rebuild the model to run inference with updated ones."""
        if getattr(self, "_train_params", None) is None:
            self._train_params = {k: torch.as_tensor(v).detach().float().to(self.device).clone().requires_grad_(True)
                                  for k, v in self._head_state.items()}
        return self._train_params

    # ------------------------------------------------------------------------------------ detection
    def _raw_input(self, batched_inputs, out=None):
        """The step's frames as ONE device tensor (`out` = a static buffer to fill, for graph replay) + how to
        normalise it: ("u8", net hw, flip) for the device ingest of SURVEY §8-f2 -- `frame_u8` (u8 [H0,W0,3] as read
        from disk) + `resize_hw` (+ `flip_channels`) -- or ("f32", hw, None) for the reference's `image` (f32 [3,H,W],
        already resized).  gom_lstmatcher.py:164-170 for same-size frames (no padding needed)."""
        first = batched_inputs[0]
        if "frame_u8" in first:
            hw, flip = tuple(first["resize_hw"]), bool(first.get("flip_channels", False))
            frames = [x["frame_u8"] for x in batched_inputs]
            for x in batched_inputs:
                if tuple(x["resize_hw"]) != hw or tuple(x["frame_u8"].shape) != tuple(frames[0].shape) \
                        or bool(x.get("flip_channels", False)) != flip:
                    raise ValueError("frames of one step must share source size, target size and channel order")
            kind, dtype = ("u8", hw, flip), torch.uint8
        else:
            frames = [x["image"] for x in batched_inputs]
            hw = tuple(frames[0].shape[-2:])
            for im in frames:
                if tuple(im.shape[-2:]) != hw:
                    raise ValueError("frames of one step must share a size (got %s and %s)" % (hw, tuple(im.shape[-2:])))
            kind, dtype = ("f32", hw, None), _f32
        if out is None:
            out = torch.empty((len(frames),) + tuple(frames[0].shape), dtype=dtype, device=self.device)
        if all(f.device == self.device and f.dtype == dtype for f in frames):
            torch.stack(frames, out=out)                         # one launch for the whole step
        elif all(f.device.type == "cpu" and f.dtype == dtype for f in frames):
            out.copy_(self._upload(frames, dtype))               # H2D on the upload stream, then one D2D launch
            self._stage_release()
        else:
            for i, f in enumerate(frames):
                out[i].copy_(f, non_blocking=True)
        return out, kind

    def _upload(self, frames, dtype):
        """Host frames (the reference hands `image` over as CPU tensors and moves them inside the timed window,
        gom_lstmatcher.py:164-170) -> one of two device staging buffers, copied on a stream of their own: the caller queues
        step i+1 while the detector of step i still runs, so the PCIe transfer hides under it instead of sitting in front of
        the detector on its stream.  Asynchronous for pinned frames; pageable ones make the host wait for each copy.
        The current stream is made to wait for the transfer; `_stage_release` marks the buffer reusable."""
        key = (len(frames), tuple(frames[0].shape), dtype)
        st = getattr(self, "_stage", None)
        if st is None or st["key"] != key:
            if st is not None:
                torch.cuda.synchronize(self.device)              # the old buffers may still be read or written
            st = {"key": key, "next": 0, "free": [None, None],
                  "buf": [torch.empty((len(frames),) + tuple(frames[0].shape), dtype=dtype, device=self.device)
                          for _ in range(2)]}
            self._stage = st
        if getattr(self, "_up_stream", None) is None:
            self._up_stream = torch.cuda.Stream(device=self.device)
        k = st["next"]
        st["next"] = k ^ 1
        st["last"] = k
        up, cur = self._up_stream, torch.cuda.current_stream()
        if st["free"][k] is not None:
            up.wait_event(st["free"][k])                         # whoever read this buffer's previous frames is done
        else:
            up.wait_stream(cur)                                  # first use: the allocation itself is stream-ordered
        with torch.cuda.stream(up):
            for i, f in enumerate(frames):
                st["buf"][k][i].copy_(f, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(up)
        cur.wait_event(ev)
        return st["buf"][k]

    def _stage_release(self):
        st = self._stage
        ev = torch.cuda.Event()
        ev.record()
        st["free"][st["last"]] = ev

    def _normalise(self, raw, kind):
        if kind[0] == "u8":
            return ops.ingest(raw, kind[1][0], kind[1][1], self.pixel_mean, self.pixel_std, kind[2])
        return ops.preprocess(raw, self.pixel_mean, self.pixel_std)

    def preprocess_image(self, batched_inputs):
        """Normalised channels-last network input of a step + its (H, W)."""
        raw, kind = self._raw_input(batched_inputs)
        return self._normalise(raw, kind), kind[1]

    def _ensure_pool(self, extra_rows):
        need = self._pool_used + extra_rows
        if self._pool is None or need > self._pool.shape[0]:
            cap = max(need, 2 * (self._pool.shape[0] if self._pool is not None else 0), 1024)
            new = torch.empty((cap, self.roi_heads.feature_dim), dtype=_f32, device=self.device)
            if self._pool is not None and self._pool_used:
                new[:self._pool_used].copy_(self._pool[:self._pool_used])
            self._pool = new

    def inference(self, batched_inputs, time_cost):
        """Detection + re-id embedding for a step of frames (gom_lstmatcher.py:268-351), batched."""
        return self.detect_finish(self.detect_launch(batched_inputs, time_cost), time_cost)

    def _detect_core(self, raw, kind, time_cost, detector=None, flag=None):
        """Every detector kernel of a step, queued on the current stream: no host sync, no host<->device copy, shapes
        fixed by (B, input kind) -- which is what lets `_detect_graphed` capture it."""
        sync = torch.cuda.synchronize if time_cost.get("_sync") else (lambda: None)
        hw = kind[1]
        t0 = time.time()
        x = self._normalise(raw, kind)
        sync(); time_cost["pre_process"] += time.time() - t0
        t0 = time.time()
        # no batch padding: the reference hands `ImageList.from_tensors(images)` (gom_lstmatcher.py:169, no
        # size_divisibility) to the backbone, and Swin pads inside its own blocks (swin_transformer.py:251-253, :320,
        # :477-479) -- so do csrc/swin.hip's patchify / window / merge kernels
        backbone, transformer = detector if detector is not None else (self.backbone, self.detection_transformer)
        feats = backbone.forward(x)
        sync(); time_cost["backbone"] += time.time() - t0
        t0 = time.time()
        out = transformer.forward([feats[k] for k in self.feature_names])
        if ops.GEMM_MODE != "f16x3":
            # the bf16x6 / exact-fp32 GEMM kernels carry no range flag: check the head outputs (a non-finite activation anywhere
            # upstream reaches them through the last layer's products) -- `detect_finish` raises instead of returning garbage
            for name in ("pred_logits", "pred_text_logits", "pred_ctrl_points", "pred_bd_points"):
                ops.flag_nonfinite(out[name], flag if flag is not None else ops.range_flag(self.device))
        sync(); time_cost["detector"] += time.time() - t0
        re = None
        if self.with_rescore:
            t0 = time.time()
            re = self.roi_heads.rescoring_head(out["query_features"])
            sync(); time_cost["rescore"] += time.time() - t0
        T = self.cfg.MODEL.TRANSFORMER
        B, nq, P = raw.shape[0], T.NUM_QUERIES, T.NUM_POINTS
        recs = ops.argmax_rows(out["pred_text_logits"])
        det = ops.detect_post(out["pred_logits"], re, out["pred_ctrl_points"], out["pred_bd_points"], recs, B, nq, P,
                              hw[0], hw[1], self.test_score_threshold, self.nms_thresh,
                              self.roi_heads.asso_thresh_test)
        return out["query_features"], det

    def _detect_graphed(self, batched_inputs, time_cost):
        """hipGraph replay of `_detect_core` (~870 launches per 8-frame step become one graph launch: the host is
        free for the tracker of the previous step, and the in-stream launch gaps disappear).  Per (B, input kind): the
        first call runs eagerly (warms every per-resolution cache), the second captures, later ones replay.  Outputs
        are copied out of the graph's static buffers, so a replay never overwrites what an unfinished step reads.
        Returns None when this step is not (yet) graphed."""
        B = len(batched_inputs)
        first = batched_inputs[0]
        src = first["frame_u8"] if "frame_u8" in first else first["image"]
        key = (B, tuple(src.shape), str(src.dtype), tuple(first.get("resize_hw", ())), bool(first.get("flip_channels")))
        state = self._graphs.pop(key, None)
        if state is not None:
            self._graphs[key] = state                           # most recently used last
        if state is None:
            self._graphs[key] = "warm"                          # this call: eager
            for k in [k for k, v in self._graphs.items() if v == "warm"][:-8]:
                del self._graphs[k]
            return None
        if time_cost.get("_sync"):
            return None                                         # per-stage timing needs the eager path
        if state == "warm":
            # a captured graph pins its whole activation pool (several GB at full size): keep the few most recently used
            # step shapes only (a video has one resolution; the next video may bring another)
            captured = [k for k, v in self._graphs.items() if isinstance(v, dict)]
            for k in captured[:max(0, len(captured) - (self.max_graphs - 1))]:
                self._graphs[k] = "warm"                        # dropped: re-captured if it comes back
            try:
                raw, kind = self._raw_input(batched_inputs)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                tc = {k: 0.0 for k in time_cost if k != "_sync"}
                # thread_local: API calls of other threads (RCCL's watchdog polling its events) must not void the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    qf, det = self._detect_core(raw, kind, tc)
                state = {"graph": g, "raw": raw, "kind": kind, "qf": qf, "det": det}
                self._graphs[key] = state
            except Exception as e:                              # capture unsupported here: stay eager, loudly
                import warnings
                warnings.warn("hipGraph capture of the detector failed (%s: %s); running eagerly" % (type(e).__name__, e))
                self.use_graphs = False
                return None
        t0 = time.time()
        self._raw_input(batched_inputs, out=state["raw"])
        state["graph"].replay()
        det = {k: (v.clone() if isinstance(v, torch.Tensor) and k in ("small", "ctrl", "bd", "recs") else v)
               for k, v in state["det"].items()}
        small = det["small"]
        o1, o2, o3 = det["small_layout"]
        nq = self.cfg.MODEL.TRANSFORMER.NUM_QUERIES
        det["count"], det["keep_idx"] = small[:o1], small[o1:o2].view(B, nq)
        det["scores"], det["boxes"] = small[o2:o3].view(_f32).view(B, nq), small[o3:].view(_f32).view(B, nq, 4)
        qf = state["qf"].clone()
        time_cost["detector"] += time.time() - t0
        return qf, det, state["kind"][1]

    def detect_launch(self, batched_inputs, time_cost):
        """Asynchronous half of `inference`: queues every detector kernel of the step on the current stream, ends
        with a non-blocking D2H copy of the nq-padded detection summary and an event.  No host sync, no tracker
        state touched -- a caller may queue the next step's detection before finishing this one."""
        assert not self.training
        lane = getattr(self, "_det_stream", None)
        if lane is not None and torch.cuda.current_stream() != lane:     # CU-partitioned step: the detector's own lane
            lane.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(lane):
                return self.detect_launch(batched_inputs, time_cost)
        B = len(batched_inputs)
        graphed = self._detect_graphed(batched_inputs, time_cost) if self.use_graphs else None
        if graphed is not None:
            qf, det, hw = graphed
        else:
            t0 = time.time()
            raw, kind = self._raw_input(batched_inputs)
            time_cost["pre_process"] += time.time() - t0
            qf, det = self._detect_core(raw, kind, time_cost)
            hw = kind[1]
        host = torch.empty((det["small"].numel() + 1,), dtype=torch.int32, pin_memory=True)
        host[:-1].copy_(det["small"], non_blocking=True)
        flag = ops.range_flag(self.device)
        host[-1:].copy_(flag, non_blocking=True)                 # f16x3 kernels: "a result was not finite"
        flag.zero_()                                             # in stream order: the next step starts with a clean flag
        ev = torch.cuda.Event()
        ev.record()
        return {"query_features": qf, "det": det, "host": host, "event": ev, "B": B, "hw": hw, "inputs": batched_inputs}

    def _fallback_detect(self, h, time_cost):
        """Re-run the detector of step `h` on the bf16x6 twin (eagerly, on the current stream) and put its results in the
        handle.  The twin is what a model built under ops.GEMM_MODE = "bf16x6" holds, so the step's results are those of a
        pure-bf16x6 run (tests/test_model_gpu.py)."""
        import warnings
        warnings.warn("gomatching_amd: an activation left fp16's range (|x| > 65504) under the f16x3 back-end; this step is "
                      "re-run on the bf16x6 kernels (fp32's exponent range, ~1.6x slower).  Build the model under "
                      "ops.GEMM_MODE = 'bf16x6' if this checkpoint does it on every step.")
        with ops.gemm_mode("bf16x6"):
            if self._fallback_det is None:
                self._fallback_det = self._build_detector(self._sd)
            raw, kind = self._raw_input(h["inputs"])
            tc = {k: 0.0 for k in time_cost if k != "_sync"}
            # a flag word of the twin's own: the next step's f16x3 detector may be running (and flagging) on the other stream
            if getattr(self, "_fb_flag", None) is None:
                self._fb_flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
            flag = self._fb_flag
            flag.zero_()
            qf, det = self._detect_core(raw, kind, tc, detector=self._fallback_det, flag=flag)
        host = torch.empty((det["small"].numel() + 1,), dtype=torch.int32, pin_memory=True)
        host[:-1].copy_(det["small"], non_blocking=True)
        host[-1:].copy_(flag, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self.fallback_steps += 1
        h.update(query_features=qf, det=det, host=host, hw=kind[1])
        if int(host[-1]) != 0:
            raise ops._lib_mod.GomError("the detector produced a non-finite value under the bf16x6 back-end as well: the input "
                                        "or the weights are not finite, or an activation overflowed fp32")

    def detect_finish(self, h, time_cost):
        """Second half: wait for the step's event (the one host sync of the step), embed the kept detections
        (FCHead4Query into the re-id pool) and build the per-frame Instances.  Runs on the CURRENT stream, which
        may differ from the one `detect_launch` used (it is made to wait on the event)."""
        det, B, hw = h["det"], h["B"], h["hw"]
        T = self.cfg.MODEL.TRANSFORMER
        nq, P = T.NUM_QUERIES, T.NUM_POINTS
        t0 = time.time()
        cur = torch.cuda.current_stream()
        cur.wait_event(h["event"])
        h["event"].synchronize()
        for tns in (h["query_features"], det["small"], det["ctrl"], det["bd"], det["recs"]):
            tns.record_stream(cur)
        if h["host"][-1] != 0:                                   # never a silent wrong result (gemm_f16x3.hip)
            if self.gemm_mode == "f16x3" and self.precision_fallback and h.get("inputs") is not None:
                self._fallback_detect(h, time_cost)
                det = h["det"]
                for tns in (h["query_features"], det["small"], det["ctrl"], det["bd"], det["recs"]):
                    tns.record_stream(cur)
            elif self.gemm_mode == "f16x3":
                raise ops._lib_mod.GomError("f16x3 GEMM produced a non-finite value: an activation left fp16's range "
                                            "(|x| > 65504) or the input was not finite; set precision_fallback or build the "
                                            "model under ops.GEMM_MODE = 'bf16x6'")
            else:
                raise ops._lib_mod.GomError("the detector produced a non-finite value (%s back-end): the input or the weights "
                                            "are not finite, or an activation overflowed fp32" % self.gemm_mode)
        small = h["host"].numpy()[:-1]
        o1, o2, o3 = det["small_layout"]
        counts = small[:o1]
        keep = small[o1:o2].reshape(B, nq)
        scores_h = small[o2:o3].view(np.float32).reshape(B, nq)
        boxes_h = small[o3:].view(np.float32).reshape(B, nq, 4)
        out = {"query_features": h["query_features"]}
        n_total = int(counts.sum())
        rows = np.concatenate([keep[b, :counts[b]] for b in range(B)]) if n_total else np.zeros((0,), np.int32)
        self._ensure_pool(n_total)
        row0 = self._pool_used
        if n_total:
            rows_d = self._h2d(rows.astype(np.int32))
            qf = out["query_features"].view(B * nq, P * T.HIDDEN_DIM)
            # split-K always: a row's embedding then never depends on how many detections share the launch
            x = ops.gemm(qf, self.roi_heads.fcs[0][0], bias=self.roi_heads.fcs[0][1], rows=rows_d, relu=True,
                         splitk=True)
            for i, (w, b) in enumerate(self.roi_heads.fcs[1:]):
                last = i == len(self.roi_heads.fcs) - 2
                x = ops.gemm(x, w, bias=b, relu=True, out=self._pool[row0:row0 + n_total] if last else None,
                             splitk=True)
            if len(self.roi_heads.fcs) == 1:
                self._pool[row0:row0 + n_total].copy_(x)
        self._pool_used += n_total
        time_cost["tracker"] += time.time() - t0
        results, off = [], row0
        for b in range(B):
            n = int(counts[b])
            r = Instances(hw)
            r.reid_features = self._pool[off:off + n]
            r.pred_boxes = Boxes(det["boxes"][b, :n])
            r.scores = det["scores"][b, :n]
            r.pred_classes = torch.zeros((n,), dtype=torch.int64, device=self.device)
            r.ctrl_points = det["ctrl"][b, :n]
            r.recs = det["recs"][b, :n]
            r.bd = det["bd"][b, :n]
            r._gom = {"boxes": boxes_h[b, :n].copy(), "scores": scores_h[b, :n].copy(), "row0": off, "ids": None,
                      "det": det, "b": b, "pool": self._pool}       # lets dist.pack_records pack the step in one launch
            results.append(r)
            off += n
        return results

    # ------------------------------------------------------------------------------------ tracking
    @staticmethod
    def _host(inst):
        """Host-side mirror (boxes, ids, pool rows) of an Instances; rebuilt for foreign Instances."""
        g = getattr(inst, "_gom", None)
        if g is None:
            g = {"boxes": inst.pred_boxes.tensor.detach().cpu().numpy(), "row0": None, "ids": None}
            inst._gom = g
        if g["ids"] is None and inst.has("track_ids"):
            g["ids"] = inst.track_ids.detach().cpu().numpy().astype(np.int64)
        return g

    def _nboxes(self, inst, hw=None):
        """Host boxes of a frame divided by an image size, as lstmatcher.py:478-494 `_get_boxes_time` divides them: by the frame's
        OWN size in a short-term match (hw None), by the size of the window's FIRST frame in a long-term match (the reference
        builds every window Instances with `full_instances[0].image_size`, gom_lstmatcher.py:471) -- which only differ when a clip
        mixes resolutions (BASELINE config #5).  The tracker kernels then run with an image size of 1 x 1 (x / 1.0f is exact), so
        the fp32 quotient is formed once, here, as the reference forms it."""
        g = self._host(inst)
        if hw is not None and tuple(hw) != tuple(inst.image_size):
            h, w = hw
            return g["boxes"].reshape(-1, 4).astype(np.float32) / np.array([w, h, w, h], np.float32)
        nb = g.get("nboxes")
        if nb is None or len(nb) != len(g["boxes"]):
            h, w = inst.image_size
            nb = g["boxes"].reshape(-1, 4).astype(np.float32) / np.array([w, h, w, h], np.float32)
            g["nboxes"] = nb
        return nb

    def _set_ids(self, inst, ids):
        """Inside `track_frames` ids live on the host while the recurrence runs and `_flush_ids` uploads them once."""
        g = self._host(inst)
        g["ids"] = np.asarray(ids, dtype=np.int64)
        g["ids_dirty"] = True
        if not getattr(self, "_defer_ids", False):               # direct callers of run_*_match get device ids at once
            self._flush_ids([inst])

    def _flush_ids(self, instances):
        """One packed host->device copy of every frame whose ids changed; `track_ids` become views of it."""
        dirty = [x for x in instances if getattr(x, "_gom", None) is not None and x._gom.get("ids_dirty")]
        if not dirty:
            return
        flat = self._h2d(np.concatenate([x._gom["ids"] for x in dirty]).astype(np.int64))
        o = 0
        for x in dirty:
            n = len(x._gom["ids"])
            x._fields["track_ids"] = flat[o:o + n]
            x._gom["ids_dirty"] = False
            o += n

    def _h2d(self, arr):
        """numpy -> device through a pinned staging ring (a pageable copy costs ~40 us and stalls the stream; the
        tracker issues one per match).  A slot is rewritten only after the event recorded behind its last copy.
        The default moves the words with a kernel rather than the DMA engine (no DMA-to-kernel hand-over, same cost): a
        hardening step after a rare run-to-run difference of the track ids was observed while the detector's hipGraph
        replayed on the other stream -- not a proven fix, root cause not established ("Tracker determinism")."""
        arr = np.ascontiguousarray(arr)
        t = torch.from_numpy(arr)
        nbytes = arr.nbytes
        if nbytes == 0:
            return t.to(self.device)
        ring = getattr(self, "_pin_ring", None)
        if ring is None or ring[0][0].numel() < nbytes:
            if ring is not None:                                 # the copy KERNELS still reading the old slots are invisible
                for _, ev_old in ring:                           # to torch's host allocator: let them finish before the
                    if ev_old is not None:                       # blocks go back to its free list
                        ev_old.synchronize()
            cap = max(1 << 16, 2 * nbytes)
            ring = [[torch.empty((cap,), dtype=torch.uint8, pin_memory=True), None] for _ in range(4)]
            self._pin_ring, self._pin_next = ring, 0
        slot = ring[self._pin_next]
        self._pin_next = (self._pin_next + 1) % len(ring)
        if slot[1] is not None:
            slot[1].synchronize()
        host = slot[0][:nbytes].view(t.dtype).view(t.shape)
        host.copy_(t)
        if self.h2d_mode == "kernel" and nbytes % 4 == 0:
            dev = ops.copy_words(host, torch.empty(t.shape, dtype=t.dtype, device=self.device))
        else:
            dev = host.to(self.device, non_blocking=True)
            if self.h2d_mode == "sync":
                torch.cuda.current_stream().synchronize()
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return dev

    def _d2h(self, t):
        """device tensor -> numpy through a pinned landing ring + one event wait: the tracker's scores never travel through
        the runtime's pageable staging (which it would share with whatever pageable upload the caller's frames are doing on
        the detector stream at that moment)."""
        if not self.pinned_d2h:
            return t.cpu().numpy()
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return t.cpu().numpy()
        ring = getattr(self, "_land_ring", None)
        if ring is None or ring[0].numel() < nbytes:
            cap = max(1 << 16, 2 * nbytes)
            ring = [torch.empty((cap,), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._land_ring, self._land_next = ring, 0
        slot = ring[self._land_next]
        self._land_next = (self._land_next + 1) % len(ring)
        host = slot[:nbytes].view(t.dtype).view(t.shape)
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ev.synchronize()
        return host.numpy().copy()

    def _reid_rows(self, inst, sel):
        """Pool rows of the selected detections of one frame (re-homes foreign features into the pool)."""
        g = self._host(inst)
        if g["row0"] is None or not self._in_pool(inst):
            n = len(inst)
            self._ensure_pool(n)
            g["row0"] = self._pool_used
            if n:
                self._pool[g["row0"]:g["row0"] + n].copy_(inst.reid_features)
            inst._fields["reid_features"] = self._pool[g["row0"]:g["row0"] + n]
            self._pool_used += n
        return g["row0"] + np.nonzero(sel)[0]

    def _home_features(self, frames):
        """Move the embeddings of every frame that still lives outside the pool (records unpacked from the all-gather)
        into it with ONE batched copy instead of one per frame."""
        foreign = [x for x in frames if len(x) and x.has("reid_features")
                   and (self._host(x)["row0"] is None or not self._in_pool(x))]
        if not foreign:
            return
        total = sum(len(x) for x in foreign)
        self._ensure_pool(total)
        r0 = self._pool_used
        torch.cat([x.reid_features for x in foreign], out=self._pool[r0:r0 + total])
        for x in foreign:
            n = len(x)
            self._host(x)["row0"] = r0
            x._fields["reid_features"] = self._pool[r0:r0 + n]
            r0 += n
        self._pool_used = r0

    def _in_pool(self, inst):
        if self._pool is None or not inst.has("reid_features"):
            return False
        f = inst.reid_features
        if f.numel() == 0:
            return True
        lo = self._pool.data_ptr()
        return lo <= f.data_ptr() < lo + self._pool.numel() * 4

    def _rows_full(self, inst):
        """Pool rows of every detection of a frame (homes foreign features first); cached on the host mirror."""
        g = self._host(inst)
        if g["row0"] is None or not self._in_pool(inst) or g.get("rows") is None or g["rows_row0"] != g["row0"]:
            self._reid_rows(inst, np.ones((len(inst),), bool))
            g["rows"], g["rows_row0"] = g["row0"] + np.arange(len(inst), dtype=np.int64), g["row0"]
        return g["rows"]

    def _match(self, window, sels, k, short_term, hw):
        """Shared arithmetic of run_short_term_match / run_long_term_match: returns (traj [n_k,M] numpy, unique ids
        [M], ids of the non-k selected detections).  `sels`: per-frame boolean masks, or their concatenation.  The
        bookkeeping runs on the window's concatenated host arrays (a handful of numpy calls whatever the window length)."""
        prof = getattr(self, "_match_prof", None)                # diagnostic (tools/match_breakdown.py): seconds per phase
        t_in = time.perf_counter() if prof is not None else 0.0
        T = len(window)
        hosts = [self._host(w) for w in window]
        lens = [len(w) for w in window]
        sel_all = np.concatenate(sels) if isinstance(sels, (list, tuple)) else sels
        sel_idx = np.nonzero(sel_all)[0]
        f_sel = np.repeat(np.arange(T), lens)[sel_idx]           # frame of every selected detection (frame-major order)
        n_arr = np.bincount(f_sel, minlength=T)
        n_t = [int(v) for v in n_arr]
        N, n_k = len(sel_idx), n_t[k]
        not_k = f_sel != k
        ids_sel = np.concatenate([h["ids"] if h["ids"] is not None else np.full((n,), -1, np.int64)   # frame k may be
                                  for h, n in zip(hosts, lens)])[sel_idx] if N else np.zeros((0,), np.int64)  # unlabelled yet
        ids = ids_sel[not_k]
        Np = N - n_k
        uniq = np.unique(ids)
        M = len(uniq)
        if n_k == 0 or M == 0:
            return np.zeros((n_k, M), np.float32), uniq, ids
        rows = np.concatenate([self._rows_full(w) for w in window])[sel_idx].astype(np.int32)
        norm_hw = None if short_term else window[0].image_size   # long-term: the window's first frame's size for every frame
        boxes = np.concatenate([self._nboxes(w, norm_hw) for w in window])[sel_idx].astype(np.float32)
        hw = (1.0, 1.0)                                          # boxes are normalised already (_nboxes)
        nonk = np.nonzero(not_k)[0]
        k_inds = np.nonzero(~not_k)[0]
        col_of = np.searchsorted(uniq, ids)
        # "last box of a track": arg-max over onehot*arange, first index on ties (gom_lstmatcher.py:436-438)
        last = np.zeros((M,), np.int64)
        if Np > 1:
            np.maximum.at(last, col_of[1:], np.arange(1, Np))   # index 0 never overrides the initial 0 (first on ties)
        meta = np.concatenate([nonk, col_of, last, k_inds]).astype(np.int32)
        offs = np.zeros((T + 1,), np.int32)
        np.cumsum(n_arr, out=offs[1:])
        dec = None
        if (not short_term) and self.decay_time > 0:
            dec = np.power(np.float32(self.decay_time), (T - 2 - f_sel[not_k]).astype(np.float32)).astype(np.float32)
        # one packed host->device copy per match: rows | frame offsets | meta | boxes (f32 bits) | decay (f32 bits)
        parts = [rows, offs, meta, boxes.reshape(-1).view(np.int32)] + ([dec.view(np.int32)] if dec is not None else [])
        t_host = time.perf_counter() if prof is not None else 0.0
        buf = self._h2d(np.concatenate(parts))
        o0 = len(rows); o1 = o0 + len(offs); o2 = o1 + len(meta); o3 = o2 + 4 * N
        traj = self.roi_heads.match_scores(self._pool, buf[:o0], buf[o0:o1], buf[o1:o2], buf[o2:o3].view(torch.float32),
                                           buf[o3:].view(torch.float32) if dec is not None else None, n_t, k,
                                           short_term, hw, M, self.with_iou,
                                           self.max_center_dist if not short_term else 0.0)
        if prof is None:
            return self._d2h(traj), uniq, ids
        t_issue = time.perf_counter()
        out = self._d2h(traj)
        t_done = time.perf_counter()
        prof["host_prep"] += t_host - t_in
        prof["issue"] += t_issue - t_host
        prof["wait"] += t_done - t_issue
        prof["matches"] += 1
        prof["rows"] += N
        return out, uniq, ids

    def _assign(self, traj, uniq, ids_nonk, n_k):
        """LSA on -traj + thresholding (gom_lstmatcher.py:447-453 / 549-555)."""
        track_ids = np.full((n_k,), -1, np.int64)
        if traj.size:
            mi, mj = ops.linear_sum_assignment(-traj.astype(np.float64))
            for i, j in zip(mi, mj):
                thresh = self.overlap_thresh if self.not_mult_thresh else \
                    self.overlap_thresh * float((ids_nonk == uniq[j]).sum())
                if traj[i, j] > np.float32(thresh):
                    track_ids[i] = uniq[j]
        return track_ids

    def precompute_short_term(self, frames, only=None):
        """Id-independent device work of every consecutive-frame short-term match of `frames`, batched
        (roi_heads.short_term_scores); one D2H for all pairs.  Returns {index of cur frame: S numpy}.  `only`: the indices of the
        cur frames to evaluate (a rank's share of a sharded clip, dist.exchange_and_track); None = every pair."""
        pairs, rows, boxes, which, off = [], [], [], [], 0
        for t in range(1, len(frames)):
            if only is not None and t not in only:
                continue
            n_prev, n_cur = len(frames[t - 1]), len(frames[t])
            if n_prev == 0 or n_cur == 0:
                continue
            for fr in (frames[t - 1], frames[t]):
                rows.append(self._reid_rows(fr, np.ones((len(fr),), bool)))
                boxes.append(self._nboxes(fr))
            pairs.append((off, n_prev, n_cur))
            which.append(t)
            off += n_prev + n_cur
        if not pairs:
            return {}
        rows_d = self._h2d(np.concatenate(rows).astype(np.int32))
        boxes_d = self._h2d(np.concatenate(boxes).astype(np.float32))
        src_all = ops.gather_rows(self._pool, rows_d)
        check = os.environ.get("GOM_TRACKER_DOUBLE_CHECK") == "1"
        tap = check and os.environ.get("GOM_TRACKER_TAP") == "1"
        if check:
            from . import roi_heads as _rh
            src_first, rows_first, boxes_first = src_all.clone(), rows_d.clone(), boxes_d.clone()
            if tap:
                _rh.TAP = tap1 = [("src", src_all), ("rows", rows_d), ("boxes", boxes_d)]   # references: no extra kernels
        scores = self.roi_heads.short_term_scores(src_all, pairs, boxes_d, (1.0, 1.0), h2d=self._h2d)
        flat = self._d2h(torch.cat([s.reshape(-1) for s in scores]))         # the one sync of the short-term path
        if check:                                                # diagnostic: were the inputs final when they were first read?
            _rh.TAP = None
            torch.cuda.current_stream().synchronize()
            again = ops.gather_rows(self._pool, rows_d)
            for name, a, b in (("SRC", src_first, again), ("ROWS", rows_first, rows_d), ("BOXES", boxes_first, boxes_d),
                               ("SRC-vs-its-clone", src_first, src_all)):
                if not torch.equal(a, b):
                    print("%s MISMATCH in precompute_short_term: %d elements differ" % (name, int((a != b).sum())), flush=True)
            if tap:
                _rh.TAP = tap2 = [("src", again), ("rows", rows_d), ("boxes", boxes_d)]
            flat2 = self._d2h(torch.cat([s.reshape(-1) for s in self.roi_heads.short_term_scores(
                again, pairs, boxes_d, (1.0, 1.0), h2d=self._h2d)]))
            _rh.TAP = None
            if not np.isfinite(flat).all():
                print("S of the FIRST evaluation is not finite: %d of %d elements" % (int((~np.isfinite(flat)).sum()), flat.size), flush=True)
            if not np.array_equal(flat, flat2):
                bad = np.nonzero(flat != flat2)[0]
                where, o = [], 0
                for t, (_, n_prev, n_cur) in zip(which, pairs):
                    b = bad[(bad >= o) & (bad < o + n_cur * n_prev)] - o
                    if len(b):
                        where.append("pair@%d %dx%d rows %s cols %s" % (t, n_cur, n_prev, sorted(set((b // n_prev).tolist())),
                                                                       sorted(set((b % n_prev).tolist()))))
                    o += n_cur * n_prev
                print("S-RECOMPUTE MISMATCH (same inputs, after a sync): max |d| %.3e, %d of %d elements; %s; pairs %s" % (
                    float(np.abs(flat - flat2).max()), len(bad), flat.size, " | ".join(where), pairs), flush=True)
                if tap:
                    torch.cuda.synchronize()
                    for (na, a), (nb, b) in zip(tap1, tap2):
                        a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
                        if not torch.equal(a2, b2):
                            ne = a2 != b2
                            rws = torch.nonzero(ne.any(1)).flatten().tolist()
                            cls = torch.nonzero(ne.any(0)).flatten().tolist()
                            print("  TAP %s %s: %d elements differ, max |d| %.3e, rows %s, cols %s%s" % (
                                na, tuple(a.shape), int(ne.sum()), float((a2.double() - b2.double()).abs().max()), rws[:24],
                                cls[:16], " ..." if len(cls) > 16 else ""), flush=True)
            elif tap:
                print("S-RECOMPUTE SAME", flush=True)
        out, o = {}, 0
        for t, (_, n_prev, n_cur) in zip(which, pairs):
            out[t] = flat[o:o + n_cur * n_prev].reshape(n_cur, n_prev)
            o += n_cur * n_prev
        return out

    def run_short_term_match(self, instances, id_count=None, S=None):
        """gom_lstmatcher.py:405-465.  `S` (from precompute_short_term) makes this a pure host step."""
        prev, cur = instances
        if S is not None or len(prev) == 0 or len(cur) == 0:
            ids_prev = self._host(prev)["ids"]
            uniq = np.unique(ids_prev)
            if S is None:
                traj = np.zeros((len(cur), len(uniq)), np.float32)
            else:
                order = np.argsort(ids_prev, kind="stable")         # column m <-> the detection carrying uniq[m]
                traj = S[:, order]
        else:
            sels = [np.ones((len(prev),), bool), np.ones((len(cur),), bool)]
            traj, uniq, _ = self._match(instances, sels, 1, True, cur.image_size)
        track_ids = self._assign(traj, uniq, self._host(prev)["ids"], len(cur))
        if id_count:
            for i in range(len(cur)):
                if track_ids[i] < 0:
                    id_count = id_count + 1
                    track_ids[i] = id_count
        self._set_ids(cur, track_ids)
        if id_count:
            return instances, id_count
        return instances, np.unique(track_ids)

    def run_long_term_match(self, full_instances, k, id_count, cur_id):
        """gom_lstmatcher.py:467-564: past frames contribute the detections of tracks that are NOT among the current
        frame's matched ids, the current frame its unmatched (-1) detections."""
        cur = np.asarray(cur_id).reshape(-1)
        hosts = [self._host(p) for p in full_instances]
        ids_all = np.concatenate([h["ids"] for h in hosts])
        n_last = len(hosts[-1]["ids"])
        sel_all = ~np.isin(ids_all, cur)
        if n_last:
            sel_all[-n_last:] = hosts[-1]["ids"] == -1
        traj, uniq, ids_nonk = self._match(full_instances, sel_all, k, False, full_instances[k].image_size)
        off_k = sum(len(h["ids"]) for h in hosts[:k])
        sel_k = sel_all[off_k:off_k + len(hosts[k]["ids"])]
        n_k = int(sel_k.sum())
        track_ids = self._assign(traj, uniq, ids_nonk, n_k)
        for i in range(n_k):
            if track_ids[i] < 0:
                id_count = id_count + 1
                track_ids[i] = id_count
        full = hosts[k]["ids"].copy()
        full[sel_k] = track_ids
        self._set_ids(full_instances[k], full)
        return full_instances, id_count

    def batch_inference(self, batched_inputs, batch_id, id_count, instances, time_cost):
        """gom_lstmatcher.py:366-403.  Detection runs per step of `frames_per_step` same-size frames on the current
        stream; the id recurrence stays strictly per frame and runs on a second (high-priority) stream, so the
        tracker of step j (host bookkeeping + small kernels + syncs) overlaps the detector of step j+1."""
        self.begin_batch(instances, len(batched_inputs))
        steps = self._steps(batched_inputs)
        if not steps:
            return instances, id_count
        main, trk = torch.cuda.current_stream(), self._tracker_stream()
        h = self.detect_launch(batched_inputs[steps[0][0]:steps[0][1]], time_cost)
        offset = 0
        for j in range(len(steps)):
            h_next = None
            if j + 1 < len(steps):
                h_next = self.detect_launch(batched_inputs[steps[j + 1][0]:steps[j + 1][1]], time_cost)
            with torch.cuda.stream(trk):
                dets = self.detect_finish(h, time_cost)
                instances, id_count = self.track_frames(dets, batch_id, id_count, instances, time_cost,
                                                        frame_offset=offset)
            offset += len(dets)
            h = h_next
        main.wait_stream(trk)
        return instances, id_count

    def _tracker_stream(self):
        if getattr(self, "_trk_stream", None) is None:
            self._trk_stream = torch.cuda.Stream(device=self.device, priority=-1)
        return self._trk_stream

    def reserve_tracker_cus(self, n_cus=32):
        """Multi-GPU steps (the replicated tracker handles world_size x 8 frames per step): give the tracker stream `n_cus`
        compute units of its own (for its per-frame recurrence) and the detector stream the rest (csrc/stream.hip).  The tracker is a recurrence of small
        dependent kernels; beside the detector each of them waits for resident workgroups to drain (0.7 ms per frame against
        0.19 ms on an idle GPU) and bounds the step from 4 GPUs on.  The first 32 mask bits are one CU per (XCD, shader engine),
        so both lanes stay symmetric over the eight XCDs (the XCD-aware tile orders of the detector's kernels keep working).
        n_cus = 0 undoes the reservation.  Not worth it on one GPU, where the tracker of 8 frames is 2 ms of a 36 ms step."""
        torch.cuda.synchronize(self.device)
        self._lane_stream = self._det_stream = None              # (the queues stay cached per mask in ops.masked_stream)
        ops._L().gom_ffn_set_stream_cus(0)
        if n_cus <= 0:
            return
        total = torch.cuda.get_device_properties(self.device).multi_processor_count
        assert 0 < n_cus < total
        words = (total + 31) // 32
        trk, det = [0] * words, [0] * words
        for i in range(total):
            (trk if i < n_cus else det)[i // 32] |= 1 << (i % 32)
        # the per-frame recurrence (gom_tracker_run's chain of small dependent launches) gets the lane; the tracker's few BIG
        # launches (embedding GEMMs of detect_finish, the batched short-term precompute) stay on the ordinary high-priority
        # tracker stream: on 32 CUs they took twice as long (bench --emulate-world 8: short-term 6.1 -> 12.7 ms per step)
        self._lane_stream = ops.masked_stream(trk, self.device)
        self._det_stream = ops.masked_stream(det, self.device)
        ops._L().gom_ffn_set_stream_cus(total - n_cus)           # the fused FFN's tail-round rule counts the detector's CUs

    def begin_batch(self, instances, num_new_frames):
        """Keep the carried-over window's embeddings addressable, drop everything older, size the pool."""
        carried = [x for x in instances[-self.test_len:] if x.has("reid_features")]
        saved = [x.reid_features.clone() for x in carried]      # may alias the pool that is about to be reused
        self._pool_used = 0
        self._proj_done = 0                                      # hoisted matcher projections follow the pool rows
        for x, f in zip(carried, saved):
            self._host(x)["row0"] = None
            x._fields["reid_features"] = f
            self._reid_rows(x, np.ones((len(x),), bool))
        self._ensure_pool(num_new_frames * self.cfg.MODEL.TRANSFORMER.NUM_QUERIES)   # no re-allocation mid-batch

    def _steps(self, batched_inputs):
        """[s0, s1) ranges of <= frames_per_step consecutive frames of ONE size (mixed-resolution clips, e.g.
        BASELINE config #5, simply start a new step at every size change)."""
        def size(x):
            if "frame_u8" in x:
                return tuple(x["frame_u8"].shape[:2]) + tuple(x["resize_hw"]) + (bool(x.get("flip_channels", False)),)
            return tuple(x["image"].shape[-2:]) if "image" in x else None

        steps, s0, n = [], 0, len(batched_inputs)
        while s0 < n:
            hw = size(batched_inputs[s0])
            s1 = s0 + 1
            while s1 < n and s1 - s0 < self.frames_per_step and size(batched_inputs[s1]) == hw:
                s1 += 1
            steps.append((s0, s1))
            s0 = s1
        return steps

    def detect_steps(self, batched_inputs, time_cost):
        dets = []
        for s0, s1 in self._steps(batched_inputs):
            dets.extend(self.inference(batched_inputs[s0:s1], time_cost))
        return dets

    def track_frames(self, dets, batch_id, id_count, instances, time_cost, frame_offset=0, st=None):
        """The per-frame id recurrence of gom_lstmatcher.py:369-403 over already detected frames
        (`frame_offset` = number of this batch's frames already tracked by earlier calls).  `st`: the short-term score matrices of
        `precompute_short_term(([instances[-1]] if instances else []) + dets)` when the caller already holds them (a sharded clip:
        every rank scores its own frame pairs and the blocks are exchanged, dist.exchange_and_track)."""
        if ops.NATIVE_TRACKER and len(dets):
            return self._track_frames_native(dets, batch_id, id_count, instances, time_cost, frame_offset, st=st)
        start_frame_id = batch_id * 100 + frame_offset
        t0 = time.time()
        base = len(instances)
        window = ([instances[-1]] if base else []) + list(dets)
        self._home_features(window)
        if st is None:
            st = self.precompute_short_term(window)              # keyed by index into `window`
        shift = 1 if base else 0
        time_cost["short_match"] += time.time() - t0
        self._defer_ids = True
        try:
            for frame_id in range(len(dets)):
                instances.append(dets[frame_id])
                real_frame_id = start_frame_id + frame_id
                S = st.get(frame_id + shift)
                if real_frame_id == 0:
                    n0 = len(instances[0])
                    self._set_ids(instances[0], np.arange(1, n0 + 1))
                    id_count = n0 + 1
                elif real_frame_id == 1:
                    t0 = time.time()
                    instances[0:2], id_count = self.run_short_term_match(instances[0:2], id_count=id_count, S=S)
                    time_cost["short_match"] += time.time() - t0
                else:
                    t0 = time.time()
                    instances[real_frame_id - 1: real_frame_id + 1], cur_id = self.run_short_term_match(
                        instances[real_frame_id - 1: real_frame_id + 1], S=S)
                    time_cost["short_match"] += time.time() - t0
                    if -1 in cur_id:
                        win_st = max(0, real_frame_id + 1 - self.test_len)
                        win_ed = real_frame_id + 1
                        t0 = time.time()
                        instances[win_st:win_ed], id_count = self.run_long_term_match(
                            instances[win_st:win_ed], k=min(self.test_len - 1, real_frame_id), id_count=id_count,
                            cur_id=cur_id)
                        time_cost["long_match"] += time.time() - t0
                ids = self._host(instances[-1])["ids"]
                assert len(ids) == len(np.unique(ids))
                if real_frame_id - self.test_len >= 0:
                    instances[real_frame_id - self.test_len].remove("reid_features")
        finally:
            self._defer_ids = False
        self._flush_ids(dets)
        return instances, id_count

    def _track_frames_native(self, dets, batch_id, id_count, instances, time_cost, frame_offset, st=None):
        """`track_frames` with the per-frame loop in native code (csrc/tracker_rt.hip): this method only gathers the host
        mirrors of the window (carried frames + new ones), runs the batched short-term device work, and hands everything
        over in ONE call; ids come back for all new frames."""
        import ctypes
        L = ops._L()
        start = batch_id * 100 + frame_offset
        t0 = time.time()
        base = len(instances)
        carried = list(instances[-max(self.test_len - 1, 1):]) if base else []
        self._home_features(carried + list(dets))
        given = st is not None
        if not given:
            st = self.precompute_short_term(([instances[-1]] if base else []) + list(dets))
        if os.environ.get("GOM_TRACKER_DOUBLE_CHECK") == "1" and not given:    # diagnostic: the same device work again, same bits?
            st2 = self.precompute_short_term(([instances[-1]] if base else []) + list(dets))
            for key in st:
                if not np.array_equal(st[key], st2[key]):
                    print("S MISMATCH pair %d (frame_offset %d): max |d| %.3e" % (key, frame_offset,
                                                                                float(np.abs(st[key] - st2[key]).max())), flush=True)
        shift = 1 if base else 0
        window = carried + list(dets)
        first_new = len(carried)
        hosts = [self._host(w) for w in window]
        n = np.asarray([len(w) for w in window], np.int32)
        tot = int(n.sum())
        boxes = np.ascontiguousarray(np.concatenate([h["boxes"].reshape(-1, 4) for h in hosts]) if tot
                                     else np.zeros((0, 4)), dtype=np.float32)
        rows = np.ascontiguousarray(np.concatenate([self._rows_full(w) for w in window]) if tot else np.zeros((0,)),
                                    dtype=np.int32)
        ids = np.full((tot,), -1, np.int64)
        o = 0
        for h, m in zip(hosts[:first_new], n[:first_new]):
            ids[o:o + m] = h["ids"]
            o += int(m)
        s_off = np.full((len(window),), -1, np.int64)
        chunks, so = [], 0
        for j in range(len(dets)):
            S = st.get(j + shift)
            if S is not None:
                s_off[first_new + j] = so
                chunks.append(np.ascontiguousarray(S, dtype=np.float32).reshape(-1))
                so += chunks[-1].size
        S_all = np.concatenate(chunks) if chunks else np.zeros((1,), np.float32)
        trk = self._native_tracker()
        self._hoist_projections(trk)
        idc = ctypes.c_long(int(id_count) if id_count else 0)
        secs = (ctypes.c_double * 2)(0.0, 0.0)
        frame_wh = np.ascontiguousarray([[w.image_size[1], w.image_size[0]] for w in window], dtype=np.float32)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        time_cost["short_match"] += time.time() - t0
        run_stream = ops._stream()
        lane = getattr(self, "_lane_stream", None)
        if lane is not None:                                     # CU-partitioned step: the recurrence runs on its own CUs
            lane.wait_stream(torch.cuda.current_stream())        # ... after the embeddings it gathers from the pool
            run_stream = lane.cuda_stream
        ops.check(L.gom_tracker_run_wh(trk, len(window), ptr(n), ptr(boxes), ptr(rows), ptr(ids), first_new, start,
                                       ptr(S_all), ptr(s_off), ctypes.c_void_p(self._pool.data_ptr()), self._pool.stride(0),
                                       ptr(frame_wh), ptr(self._decay_table), ctypes.byref(idc), secs, run_stream),
                  "gom_tracker_run_wh")
        if lane is not None:
            torch.cuda.current_stream().wait_stream(lane)        # the pool may be rewritten by later work on this stream
        time_cost["short_match"] += secs[0]
        time_cost["long_match"] += secs[1]
        self._defer_ids = True
        try:
            o = int(n[:first_new].sum())
            for j, d in enumerate(dets):
                m = int(n[first_new + j])
                self._set_ids(d, ids[o:o + m].copy())
                o += m
                instances.append(d)
                real = start + j
                if real - self.test_len >= 0:
                    instances[real - self.test_len].remove("reid_features")
        finally:
            self._defer_ids = False
        self._flush_ids(dets)
        return instances, int(idc.value)

    def _native_tracker(self):
        """The native tracker handle for the CURRENT thresholds and matcher weights.  The handle caches both at creation, so it
        is keyed on them (a changed `overlap_thresh` / `decay_time` / ..., or re-created matcher tensors, make a new handle
        and destroy the old one with its device and pinned buffers) and released by `close()`."""
        m = self.roi_heads._matcher(False)
        key = (int(self.test_len), float(self.overlap_thresh), bool(self.not_mult_thresh), float(self.decay_time),
               bool(self.with_iou), float(self.max_center_dist), bytes(m._enc_c) if len(m.enc) else b"",
               bytes(m._dec_c) if len(m.dec) else b"", m.d, m.heads, m.ffn)
        if getattr(self, "_ntrk", None) is not None and self._ntrk_key == key:
            return self._ntrk
        self.close()
        L = ops._L()
        self._ntrk = L.gom_tracker_create(self.test_len, float(self.overlap_thresh), 1 if self.not_mult_thresh else 0,
                                          1 if self.decay_time > 0 else 0, 1 if self.with_iou else 0,
                                          float(self.max_center_dist), m._enc_c, len(m.enc), m._dec_c, len(m.dec), m.d,
                                          m.heads, m.ffn)
        if not self._ntrk:
            raise ops._lib_mod.GomError("gom_tracker_create failed")
        self._ntrk_key = key
        self._decay_table = np.power(np.float32(self.decay_time if self.decay_time > 0 else 1.0),
                                     np.arange(self.test_len + 1).astype(np.float32)).astype(np.float32)
        return self._ntrk

    def _hoist_projections(self, trk):
        """The matcher's two per-row projections of the RAW embeddings (encoder layer 0 q | k | v, decoder layer 0 query) for
        every pool row that does not have them yet, handed to the next `gom_tracker_run` (gom_match_scores_proj_f32): the
        largest product of a long-term match and one more leave its chain of dependent launches -- computed here once per
        detection, for all new rows of the step at once, by the same small-GEMM kernel (same bits)."""
        m = self.roi_heads._matcher(False)
        if not ops.HOIST_MATCH_PROJECTIONS or not m.enc or not m.dec or self._pool is None:
            return
        d, L = m.d, ops._L()
        cap = self._pool.shape[0]
        if getattr(self, "_proj", None) is None or self._proj.shape[0] < cap or self._proj_pool is not self._pool:
            self._proj = torch.empty((cap, 4 * d), dtype=_f32, device=self.device)
            self._proj_pool, self._proj_done = self._pool, 0
        self._proj_done = min(getattr(self, "_proj_done", 0), self._pool_used)
        a, b = self._proj_done, self._pool_used
        if b > a:
            w_in, b_in = m.enc[0]["in"]
            w_q, b_q = m.dec[0]["in"]
            ops.gemm_small_rows(self._pool[a:b], w_in, b_in, self._proj[a:b, :3 * d])
            ops.gemm_small_rows(self._pool[a:b], w_q[:d], b_q[:d], self._proj[a:b, 3 * d:])
            self._proj_done = b
        ops.check(L.gom_tracker_set_projections(trk, self._proj.data_ptr(), self._proj.stride(0)), "gom_tracker_set_projections")

    def close(self):
        """Release the native tracker handle (hipMalloc / hipHostMalloc buffers of csrc/tracker_rt.hip)."""
        h, self._ntrk = getattr(self, "_ntrk", None), None
        if h:
            torch.cuda.synchronize(self.device)                  # its buffers may still be read by queued kernels
            ops._L().gom_tracker_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:                                        # interpreter shutdown: the library may be gone already
            pass

    def _remove_short_track(self, instances):
        """gom_lstmatcher.py:566-577.  The per-frame boolean indexing of the reference (seven index kernels and a
        host->device copy per frame) is done for all frames at once: one copy of the kept positions, then one
        cat + one index_select per field, the per-frame results being views of the selected rows."""
        ids = np.concatenate([self._host(x)["ids"] for x in instances]) if instances else np.zeros((0,), np.int64)
        uniq, counts = np.unique(ids, return_counts=True)
        short_arr = uniq[counts < self.min_track_len]
        todo, keeps, gidx, off = [], [], [], 0
        for k in range(len(instances)):
            hid = self._host(instances[k])["ids"]
            keep = ~np.isin(hid, short_arr)
            if keep.all():                                       # nothing to drop in this frame
                continue
            todo.append(k)
            keeps.append(keep)
            gidx.append(off + np.nonzero(keep)[0])
            off += len(hid)
        if not todo:
            return instances
        idx = self._h2d(np.concatenate(gidx).astype(np.int64))
        frames = [instances[k] for k in todo]
        common = [f for f in frames[0]._fields if all(x.has(f) for x in frames)]
        n_keep = [int(kp.sum()) for kp in keeps]
        picked = {}
        for f in common:
            vals = [x._fields[f] for x in frames]
            boxes = isinstance(vals[0], Boxes)
            cat = torch.cat([(v.tensor if boxes else v) for v in vals])
            sel, o, outs = cat.index_select(0, idx), 0, []
            for n in n_keep:
                outs.append(Boxes(sel[o:o + n]) if boxes else sel[o:o + n])
                o += n
            picked[f] = outs
        for j, k in enumerate(todo):
            old, keep = instances[k], keeps[j]
            new = Instances(old.image_size)
            local = None
            for f, v in old._fields.items():
                if f in picked:
                    new.set(f, picked[f][j])
                else:                                            # a field only some frames carry (reid_features)
                    if local is None:
                        local = torch.from_numpy(np.nonzero(keep)[0]).to(self.device)
                    new.set(f, v[local])
            g = old._gom
            new._gom = {"boxes": g["boxes"][keep], "ids": g["ids"][keep], "row0": None}
            instances[k] = new
        return instances

    def batch_postprocess(self, instances, image_sizes):
        """gom_lstmatcher.py:353-364 + detector_postprocess :78-111 (both branches): scale ctrl_points and
        bd to the original frame size; pred_boxes stay in network-input pixels.  Frames that share the scale factors
        (a whole video normally) are scaled by ONE kernel per field over their concatenated rows."""
        groups = {}
        for r in instances:                                      # results leave the model here: they must not alias the pool
            if r.has("reid_features") and self._in_pool(r) and len(r):
                r._fields["reid_features"] = r.reid_features.clone()
                self._host(r)["row0"] = None
        for i, (r, image_size) in enumerate(zip(instances, image_sizes)):
            sx, sy = image_size[1] / r.image_size[1], image_size[0] / r.image_size[0]
            if self.min_size_test and self.max_size_test:
                # ViTAE branch (:82-96): the network input was padded, so the scale comes from the resize rule
                # (ResizeShortestEdge re-derived from the output size), not from the padded tensor's shape
                oh, ow = image_size
                size = self.min_size_test * 1.0
                k = self.min_size_test / min(ow, oh)
                newh, neww = (size, k * ow) if oh < ow else (k * oh, size)
                if max(newh, neww) > self.max_size_test:
                    k = self.max_size_test * 1.0 / max(newh, neww)
                    newh, neww = newh * k, neww * k
                sx, sy = ow / int(neww + 0.5), oh / int(newh + 0.5)
            key = (sx, sy, r.has("ctrl_points"), r.has("pred_boxes") and not isinstance(r.bd, list))
            groups.setdefault(key, []).append(i)
        for (sx, sy, has_ctrl, has_bd), idxs in groups.items():
            for field, on in (("ctrl_points", has_ctrl), ("bd", has_bd)):
                if not on:
                    continue
                vals = [instances[i]._fields[field] for i in idxs]
                if len(vals) == 1:
                    instances[idxs[0]]._fields[field] = ops.scale_xy_(vals[0].contiguous(), sx, sy)
                    continue
                cat = ops.scale_xy_(torch.cat(vals), sx, sy)       # cat allocates: the per-frame inputs stay intact
                o = 0
                for i, v in zip(idxs, vals):
                    instances[i]._fields[field] = cat[o:o + v.shape[0]]
                    o += v.shape[0]
        return [{"instances": r} for r in instances]
