"""LST-Matcher association heads on MI355X.

Mirrors of the reference's ROI_HEADS classes, same names and inference-side methods:
  * `LSTMatcher`       /root/reference/gomatching/modeling/roi_heads/lstmatcher.py:59-557
  * `SHA_FFN_CRSATTN`  /root/reference/gomatching/modeling/roi_heads/shared_ffn_crsattn.py:62-535
Only the shipped-config shape is built (NUM_WEIGHT_LAYERS 0 -> bare dot-product predictor, NO_POS_EMB,
NORM False, NO_DECODER_SELF_ATT): other shapes raise at construction.  Training methods
(`loss_res`, `_get_asso_gt`, `detr_asso_loss`) are out of scope.
"""
import numpy as np
import torch

from .. import ops

_f32 = torch.float32

TAP = None          # diagnostic: a list that receives (stage name, tensor) of every intermediate of the short-term path


def _tap(name, t):
    if TAP is not None:
        TAP.append((name, t))
    return t


def _dev(t, device):
    return t.detach().float().contiguous().to(device)


class _MatcherTransformer:
    """roi_heads/transformer.py:19-96 with norm=False: post-norm layers whose norms are Identity."""

    def __init__(self, sd, name, device, d, heads, n_enc, n_dec, only_dec_crs_attn):
        self.d, self.heads, self.only_crs = d, heads, only_dec_crs_attn
        g = lambda k: _dev(sd[name + "." + k], device)
        self.enc = []
        for i in range(n_enc):
            p = "encoder.layers.%d." % i
            self.enc.append({"in": (g(p + "self_attn.in_proj_weight"), g(p + "self_attn.in_proj_bias")),
                             "out": (g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias")),
                             "lin1": (g(p + "linear1.weight"), g(p + "linear1.bias")),
                             "lin2": (g(p + "linear2.weight"), g(p + "linear2.bias"))})
        self.dec = []
        for i in range(n_dec):
            p = "decoder.layers.%d." % i
            L = {"in": (g(p + "multihead_attn.in_proj_weight"), g(p + "multihead_attn.in_proj_bias")),
                 "out": (g(p + "multihead_attn.out_proj.weight"), g(p + "multihead_attn.out_proj.bias"))}
            if not only_dec_crs_attn:
                L["lin1"] = (g(p + "linear1.weight"), g(p + "linear1.bias"))
                L["lin2"] = (g(p + "linear2.weight"), g(p + "linear2.bias"))
            self.dec.append(L)
        self.ffn = self.enc[0]["lin1"][0].shape[0] if self.enc else \
            (self.dec[0]["lin1"][0].shape[0] if self.dec and "lin1" in self.dec[0] else 0)
        self._enc_c, self._dec_c = ops.matcher_layers(self.enc), ops.matcher_layers(self.dec)   # native-runtime view

    def _attend(self, q, k, v, ld_q, ld_kv, Lq, Lk):
        out = torch.empty((Lq, self.d), dtype=_f32, device=q.device)
        hd = self.d // self.heads
        ops.mha_core(q, k, v, out, 1, 1, self.heads, hd, Lq, Lk,
                     [0, 0, ld_q, 0, 0, ld_kv, 0, 0, ld_kv, 0, 0, self.d])
        return out

    def forward(self, src, lo, hi):
        """src [N,F] (frames concatenated), query rows [lo,hi) -> (feats [M,F], memory [N,F])."""
        N, E = src.shape
        memory = src
        for L in self.enc:                                   # forward_post, norms = Identity (transformer.py:180-195)
            qkv = ops.gemm(memory, L["in"][0], bias=L["in"][1], small=True)                       # [N, 3E]
            f = qkv.view(-1)
            a = self._attend(f, f[E:], f[2 * E:], 3 * E, 3 * E, N, N)
            memory = ops.gemm(a, L["out"][0], bias=L["out"][1], R=memory, small=True)
            h = ops.gemm(memory, L["lin1"][0], bias=L["lin1"][1], relu=True, small=True)
            memory = ops.gemm(h, L["lin2"][0], bias=L["lin2"][1], R=memory, small=True)
        tgt = src[lo:hi]                                     # tgt = src[query_inds] (transformer.py:80-84)
        M = hi - lo
        for L in self.dec:                                   # no decoder self-attention (transformer.py:270-294)
            w, b = L["in"]
            q = ops.gemm(tgt, w[:E], bias=b[:E], small=True)
            kv = ops.gemm(memory, w[E:], bias=b[E:], small=True)                                  # [N, 2E]
            f = kv.view(-1)
            a = self._attend(q, f, f[E:], E, 2 * E, M, N)
            tgt = ops.gemm(a, L["out"][0], bias=L["out"][1], R=tgt, small=True)
            if not self.only_crs:
                h = ops.gemm(tgt, L["lin1"][0], bias=L["lin1"][1], relu=True, small=True)
                tgt = ops.gemm(h, L["lin2"][0], bias=L["lin2"][1], R=tgt, small=True)
        return tgt, memory


    def forward_pairs(self, src_all, pairs, seg=None):
        """Batched forward over independent (previous frame, current frame) pairs -- the short-term matcher input
        depends only on the two frames' embeddings, never on track ids, so all pairs of a batch of frames share
        every GEMM (weights streamed once); the attention cores run as ONE ragged launch per layer when `seg` =
        (enc segments, dec segments, cur rows, max n, max n_cur) device descriptors are given, else per pair.
        src_all [sum N_p, F]: per pair the previous frame's rows then the current frame's.
        pairs: list of (row offset, n_prev, n_cur).  Returns (tgt [sum n_cur, F], memory [sum N_p, F], cur_off)."""
        E, H = self.d, self.heads
        Nall = src_all.shape[0]
        memory = src_all
        for L in self.enc:
            qkv = _tap("enc.qkv", ops.gemm(memory, L["in"][0], bias=L["in"][1], small=True))
            a = torch.empty((Nall, E), dtype=_f32, device=src_all.device)
            if seg is not None:
                f = qkv.view(-1)
                ops.mha_core_segments(f, f[E:], f[2 * E:], a, seg[0], len(pairs), H, E // H, 3 * E, 3 * E, 3 * E, E,
                                      seg[3], seg[3])
            else:
                for off, n_prev, n_cur in pairs:
                    n = n_prev + n_cur
                    f = qkv[off:off + n].view(-1)
                    self._attend_into(a[off:off + n], f, f[E:], f[2 * E:], 3 * E, 3 * E, n, n)
            _tap("enc.attn", a)
            memory = _tap("enc.out", ops.gemm(a, L["out"][0], bias=L["out"][1], R=memory, small=True))
            h = _tap("enc.lin1", ops.gemm(memory, L["lin1"][0], bias=L["lin1"][1], relu=True, small=True))
            memory = _tap("enc.lin2", ops.gemm(h, L["lin2"][0], bias=L["lin2"][1], R=memory, small=True))
        if seg is not None:
            cur_rows = seg[2]
        else:
            cur_rows = torch.cat([torch.arange(off + n_prev, off + n_prev + n_cur, dtype=torch.int32)
                                  for off, n_prev, n_cur in pairs]).to(src_all.device)
        tgt = _tap("dec.tgt0", ops.gather_rows(src_all, cur_rows))
        M = tgt.shape[0]
        cur_off = [0]
        for _, _, n_cur in pairs:
            cur_off.append(cur_off[-1] + n_cur)
        for L in self.dec:
            w, b = L["in"]
            q = _tap("dec.q", ops.gemm(tgt, w[:E], bias=b[:E], small=True))
            kv = _tap("dec.kv", ops.gemm(memory, w[E:], bias=b[E:], small=True))
            a = torch.empty((M, E), dtype=_f32, device=src_all.device)
            if seg is not None:
                f = kv.view(-1)
                ops.mha_core_segments(q, f, f[E:], a, seg[1], len(pairs), H, E // H, E, 2 * E, 2 * E, E, seg[4], seg[3])
            else:
                for i, (off, n_prev, n_cur) in enumerate(pairs):
                    n = n_prev + n_cur
                    f = kv[off:off + n].view(-1)
                    self._attend_into(a[cur_off[i]:cur_off[i + 1]], q[cur_off[i]:cur_off[i + 1]], f, f[E:], E, 2 * E,
                                      n_cur, n)
            _tap("dec.attn", a)
            tgt = _tap("dec.out", ops.gemm(a, L["out"][0], bias=L["out"][1], R=tgt, small=True))
            if not self.only_crs:
                h = _tap("dec.lin1", ops.gemm(tgt, L["lin1"][0], bias=L["lin1"][1], relu=True, small=True))
                tgt = _tap("dec.lin2", ops.gemm(h, L["lin2"][0], bias=L["lin2"][1], R=tgt, small=True))
        return tgt, memory, cur_off

    def _attend_into(self, out, q, k, v, ld_q, ld_kv, Lq, Lk):
        hd = self.d // self.heads
        ops.mha_core(q, k, v, out, 1, 1, self.heads, hd, Lq, Lk, [0, 0, ld_q, 0, 0, ld_kv, 0, 0, ld_kv, 0, 0, self.d])


class _MatcherBase:
    def __init__(self, cfg, sd, device, prefix="roi_heads."):
        A = cfg.MODEL.ASSO_HEAD
        if A.NUM_WEIGHT_LAYERS != 0 or not A.NO_POS_EMB or A.NORM or not A.NO_DECODER_SELF_ATT or A.NO_ENCODER_SELF_ATT:
            raise NotImplementedError("association head shape outside the shipped configs (see module docstring)")
        self.cfg, self.device, self.prefix = cfg, device, prefix
        self.feature_dim = A.FC_DIM
        self.num_fc = A.NUM_FC
        self.with_rescore = cfg.MODEL.ROI_HEADS.WITH_RESR
        self.asso_thresh_train = A.ASSO_THRESH
        self.fcs = [(_dev(sd[prefix + "asso_head.fc%d.weight" % (k + 1)], device),
                     _dev(sd[prefix + "asso_head.fc%d.bias" % (k + 1)], device)) for k in range(self.num_fc)]
        if self.with_rescore:
            self._rescoring = (_dev(sd[prefix + "rescoring_head.weight"], device),
                               _dev(sd[prefix + "rescoring_head.bias"], device))

    def rescoring_head(self, query_features):
        """nn.Linear(256, 1) on every point query (lstmatcher.py:185-186; call gom_lstmatcher.py:286-289)."""
        x = query_features.reshape(-1, query_features.shape[-1])
        return ops.gemm(x, self._rescoring[0], bias=self._rescoring[1])

    def asso_head(self, query_features, rows):
        """FCHead4Query (association_head.py:100-122) on the gathered rows of [B*nq, 25*256]."""
        x = ops.gemm(query_features, self.fcs[0][0], bias=self.fcs[0][1], rows=rows, relu=True)
        for w, b in self.fcs[1:]:
            x = ops.gemm(x, w, bias=b, relu=True)
        return x

    def _matcher(self, short_term):
        raise NotImplementedError

    def _forward_transformer(self, reid_features, n_t, query_frame, short_term=False):
        """lstmatcher.py:333-370 (eval; no positional/temporal embedding): returns asso logits [n_k, N]."""
        lo, hi = sum(n_t[:query_frame]), sum(n_t[:query_frame + 1])
        feats, memory = self._matcher(short_term).forward(reid_features, lo, hi)
        return ops.gemm(feats, memory, small=True)            # ATTWeightHead with 0 layers: q . k^T

    def match_scores(self, pool, rows, frame_offsets, meta, boxes, decay, n_t, query_frame, short_term, hw, num_tracks,
                     with_iou, max_center_dist):
        """Device side of one match (gom_lstmatcher.py:405-445 / 467-547 minus the host LSA): trajectory scores
        [n_k, num_tracks] of frame `query_frame`'s selected detections.  One native call (csrc/matcher_rt.cpp), or --
        ops.NATIVE_MATCHER = False -- the same chain composed kernel by kernel from Python; both give the same bits."""
        N, T = sum(n_t), len(n_t)
        lo, hi = sum(n_t[:query_frame]), sum(n_t[:query_frame + 1])
        m = self._matcher(short_term)
        if ops.NATIVE_MATCHER:
            return ops.match_scores(pool, rows, frame_offsets, meta, boxes, decay, N, T, lo, hi, num_tracks, m._enc_c,
                                    len(m.enc), m._dec_c, len(m.dec), m.d, m.heads, m.ffn, hw[1], hw[0], with_iou,
                                    max_center_dist)
        src = ops.gather_rows(pool, rows)
        asso = self._forward_transformer(src, n_t, query_frame, short_term=short_term)
        act = ops.asso_activate(asso, frame_offsets, T)
        return ops.track_score(act, meta, decay, boxes, hw[1], hw[0], hi - lo, N - (hi - lo), num_tracks, with_iou,
                               max_center_dist)

    def short_term_scores(self, src_all, pairs, boxes_all, image_size, h2d=None):
        """For every (prev, cur) pair: S[i, j] = max(activated association of cur i with prev j, IoU(i, j)).
        Within one frame track ids are unique, so the reference's trajectory score of cur detection i for track m
        (gom_lstmatcher.py:429-445) is exactly S[i, j_m] with j_m the previous-frame detection carrying id m: the
        whole device side of short-term matching is id-independent and is done here for all pairs at once.
        boxes_all [sum N_p, 4] px, same row order as src_all.  Returns a list of device tensors [n_cur, n_prev]."""
        max_prev = max(p[1] for p in pairs)
        if ops.BATCHED_SHORT_TERM and max_prev <= ops.SHORT_TERM_MAX_PREV:
            # every pair in one launch per op: ragged attention (2 layers), fused logits + softmax + IoU
            P = len(pairs)
            seg_enc, seg_dec, desc, row_pair, cur_rows = [], [], [], [], []
            cur_off = s_off = 0
            for i, (off, n_prev, n_cur) in enumerate(pairs):
                n = n_prev + n_cur
                seg_enc += [off, n, off, n]
                seg_dec += [cur_off, n_cur, off, n]
                desc += [off, n_prev, n_cur, cur_off, off, s_off]
                row_pair.append(np.full((n_cur,), i, np.int32))
                cur_rows.append(np.arange(off + n_prev, off + n, dtype=np.int32))
                cur_off += n_cur
                s_off += n_cur * n_prev
            parts = [np.asarray(seg_enc, np.int32), np.asarray(seg_dec, np.int32), np.asarray(desc, np.int32),
                     np.concatenate(row_pair), np.concatenate(cur_rows)]
            buf = h2d(np.concatenate(parts)) if h2d is not None else torch.from_numpy(np.concatenate(parts)).to(self.device)
            _tap("desc", buf)
            o = np.cumsum([0] + [len(p) for p in parts])
            seg = (buf[o[0]:o[1]], buf[o[1]:o[2]], buf[o[4]:o[5]], max(p[1] + p[2] for p in pairs),
                   max(p[2] for p in pairs))
            tgt, memory, _ = self._matcher(True).forward_pairs(src_all, pairs, seg)
            S = ops.short_term_pairs(tgt, memory, buf[o[2]:o[3]], buf[o[3]:o[4]], boxes_all, image_size[1], image_size[0],
                                     self.cfg.VIDEO_TEST.WITH_IOU, cur_off, max_prev, s_off)
            _tap("S", S)
            out, s_off = [], 0
            for off, n_prev, n_cur in pairs:
                out.append(S[s_off:s_off + n_cur * n_prev].view(n_cur, n_prev))
                s_off += n_cur * n_prev
            return out
        tgt, memory, cur_off = self._matcher(True).forward_pairs(src_all, pairs)
        logits = [ops.gemm(tgt[cur_off[i]:cur_off[i + 1]], memory[off:off + n_prev + n_cur], small=True)
                  for i, (off, n_prev, n_cur) in enumerate(pairs)]
        # one packed host->device copy of every pair's frame offsets and (nonk | col_of | last | k) index lists
        parts, where, o = [], [], 0
        for off, n_prev, n_cur in pairs:
            ident = np.arange(n_prev, dtype=np.int32)
            p = np.concatenate([np.array([0, n_prev, n_prev + n_cur], np.int32), ident, ident, ident,
                                np.arange(n_prev, n_prev + n_cur, dtype=np.int32)])
            parts.append(p)
            where.append(o)
            o += len(p)
        buf = h2d(np.concatenate(parts)) if h2d is not None else torch.from_numpy(np.concatenate(parts)).to(self.device)
        out = []
        for (off, n_prev, n_cur), lg, o in zip(pairs, logits, where):
            act = ops.asso_activate(lg, buf[o:o + 3], 2)
            out.append(ops.track_score(act, buf[o + 3:o + 3 + 3 * n_prev + n_cur], None,
                                       boxes_all[off:off + n_prev + n_cur], image_size[1], image_size[0], n_cur, n_prev,
                                       n_prev, self.cfg.VIDEO_TEST.WITH_IOU, 0.0))
        return out

    def _activate_asso(self, asso_logits, n_t):
        """lstmatcher.py:373-381 on the concatenated [n_k, N] logits."""
        offs = [0]
        for n in n_t:
            offs.append(offs[-1] + n)
        offs = torch.tensor(offs, dtype=torch.int32).to(self.device)
        return ops.asso_activate(asso_logits, offs, len(n_t))


class LSTMatcher(_MatcherBase):
    def __init__(self, cfg, sd, device, prefix="roi_heads."):
        super().__init__(cfg, sd, device, prefix)
        A = cfg.MODEL.ASSO_HEAD
        t = A.ASSO_THRESH_TEST
        self.asso_thresh_test = t if t > 0 else A.ASSO_THRESH
        mk = lambda n: _MatcherTransformer(sd, prefix + n, device, A.FC_DIM, A.NUM_HEADS, A.NUM_ENCODER_LAYERS,
                                           A.NUM_DECODER_LAYERS, False)
        self.long_term_matcher = mk("long_term_matcher")
        self.short_term_matcher = mk("short_term_matcher")

    def _matcher(self, short_term):
        return self.short_term_matcher if short_term else self.long_term_matcher


class SHA_FFN_CRSATTN(_MatcherBase):
    def __init__(self, cfg, sd, device, prefix="roi_heads."):
        super().__init__(cfg, sd, device, prefix)
        A = cfg.MODEL.ASSO_HEAD
        t = cfg.MODEL.TRANSFORMER.INFERENCE_TH_TEST                     # shared_ffn_crsattn.py:160
        self.asso_thresh_test = t if t > 0 else A.ASSO_THRESH
        self.shared_matcher = _MatcherTransformer(sd, prefix + "shared_matcher", device, A.FC_DIM, A.NUM_HEADS, 0,
                                                  A.NUM_DECODER_LAYERS, True)

    def _matcher(self, short_term):
        return self.shared_matcher


def build_roi_heads(cfg, sd, device):
    name = cfg.MODEL.ROI_HEADS.NAME
    if name == "LSTMatcher":
        return LSTMatcher(cfg, sd, device)
    if name == "SHA_FFN_CRSATTN":
        return SHA_FFN_CRSATTN(cfg, sd, device)
    raise ValueError("unknown MODEL.ROI_HEADS.NAME %r" % name)
