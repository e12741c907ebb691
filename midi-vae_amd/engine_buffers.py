"""Static description of the recurrent layers and the device buffers of an engine (mixin of engine.Engine): which recurrences
the model has (reference vae_definition.py:443-480 encoder branches, 519-726 decoder heads), and every HBM buffer a step reads or
writes, sized for the engine's maximum batch."""
from __future__ import annotations

import torch

from . import hiplib as hl
from . import ops
from .layout import dec_init_blocks
from .slots import *        # noqa: F401,F403
from .slots import X_EXT, X_GATHER2


class _Rec(object):
    """One recurrent layer: its parameter prefix, geometry, input mode and (per batch size) its buffers."""

    def __init__(self, prefix, T, xmode, K, init_block=None, lower=None):
        self.prefix, self.T, self.xmode, self.K = prefix, T, xmode, K
        self.init_block = init_block      # first column block of dec.init holding this cell's initial state(s)
        self.lower = lower                # the layer whose h sequence feeds this one (X_DENSE)


class _Head(object):
    """One decoder head (reference vae_definition.py:521-726): a cell stack stepped ``T`` times on a constant input, a Dense with
    softmax (kind 0, N classes, categorical cross-entropy) or sigmoid (kind 1, squared error) on the top cell's output."""

    def __init__(self, name, layers, kind, N, weight, slot, target, stream=None):
        self.name, self.layers, self.kind, self.N, self.weight, self.slot, self.target = name, layers, kind, N, weight, slot, target
        self.T = layers[0].T
        self.out = "dec.%s.out" % name
        self.stream = stream              # None = the critical stream (the notes stack)
        self.NP = None


class _Aux(object):
    """A style classifier hung on a decoder head's softmax output (reference vae_definition.py:747-761): Keras RNN over the
    (T, B, N) probabilities -> Dense(C, softmax) on the last state."""

    def __init__(self, key, src, rec, weight, slot):
        self.key, self.src, self.rec, self.weight, self.slot = key, src, rec, weight, slot
        self.head = _Head(key, [rec], 0, 0, weight, slot, "in.c_idx")
        self.head.T, self.head.out = 1, key + ".out"


class Buffers(object):
    # ------------------------------------------------------------------------------------------------------
    # static description of the recurrent layers
    # ------------------------------------------------------------------------------------------------------
    def _build_graph_description(self):
        s = self.spec
        self.enc_notes, self.enc_bi = [], []
        layers = s.enc_layers()
        if len(layers) > 1 and len(layers[0]) > 1:
            # bidirectional stack: every layer a [forward, backward] pair of records, ONE plain layer on top (_enc_bi_forward)
            for li, layer in enumerate(layers):
                self.enc_bi.append([_Rec(prefix, s.T, hl.X_INDEX if li == 0 else X_EXT, k) for prefix, _, k in layer])
            self.enc_notes = [self.enc_bi[-1][0]]
        else:           # (bidirectional with Le = 2 builds no Bidirectional layer at all: one plain layer - as written)
            for l, layer in enumerate(layers):
                first = X_GATHER2 if s.attach else hl.X_INDEX        # (two-hot rows: two table rows per step, written out first)
                self.enc_notes.append(_Rec(layer[0][0], s.T, first if l == 0 else hl.X_DENSE, layer[0][2],
                                           lower=self.enc_notes[-1] if l else None))
        self.enc_instr = _Rec("enc.instr", s.V, hl.X_INDEX, s.ID) if s.meta_instrument else None
        self.enc_vel = _Rec("enc.vel", s.T, hl.X_SCALAR, 1) if s.meta_velocity else None
        self.enc_held = _Rec("enc.held", s.T, hl.X_INDEX, 2) if s.meta_held else None
        # (record, stream, input buffer): the meta rolls beside the notes stack, in the order they are concatenated
        self.enc_meta = [m for m in ((self.enc_instr, self.s_instr, "in.i_idx"), (self.enc_vel, self.s_vel, "in.vel"),
                                     (self.enc_held, self.s_held, "in.d_idx")) if m[0] is not None]
        blocks = dec_init_blocks(s)
        self.n_init = len(blocks)
        self.dec_notes = []
        for l in range(s.Ld):
            self.dec_notes.append(_Rec("dec.notes.%d" % l, s.T, hl.X_CONST if l == 0 else hl.X_DENSE,
                                       s.Dout if l == 0 else s.H, init_block=blocks.index("dec.notes.init.%d.0" % l),
                                       lower=self.dec_notes[-1] if l else None))
        self.dec_instr = (_Rec("dec.instr.cell", s.V, hl.X_CONST, s.ID, init_block=blocks.index("dec.instr.init.0"))
                          if s.meta_instrument else None)
        self.dec_vel = (_Rec("dec.vel.cell", s.T, hl.X_CONST, 1, init_block=blocks.index("dec.vel.init.0"))
                        if s.meta_velocity else None)
        self.dec_held = (_Rec("dec.held.cell", s.T, hl.X_CONST, 2, init_block=blocks.index("dec.held.init.0"))
                         if s.meta_held else None)
        self.dec_next = []
        if s.meta_next:
            for l in range(s.Ld):
                self.dec_next.append(_Rec("dec.next.%d" % l, s.T, hl.X_CONST if l == 0 else hl.X_DENSE,
                                          s.Dout if l == 0 else s.H, init_block=blocks.index("dec.next.init.%d.0" % l),
                                          lower=self.dec_next[-1] if l else None))
        # decoder heads: the notes stack on the critical stream, every other head on its own
        self.heads = [_Head("notes", self.dec_notes, 0, s.Dout, 1.0, S_NOTES_LOSS, "in.y_idx")]
        if s.meta_instrument:
            self.heads.append(_Head("instr", [self.dec_instr], 0, s.ID, s.w_instr, S_INSTR_LOSS, "in.i_idx", self.s_instr))
        if s.meta_velocity:
            self.heads.append(_Head("vel", [self.dec_vel], 1, 1, s.w_vel, S_VEL_LOSS, "in.vel", self.s_vel))
        if s.meta_held:
            self.heads.append(_Head("held", [self.dec_held], 0, 2, s.w_held, S_HELD_LOSS, "in.d_idx", self.s_held))
        if s.meta_next:
            self.heads.append(_Head("next", self.dec_next, 0, s.Dout, s.w_next, S_NEXT_LOSS, "in.n_idx", self.s_next))
        self.head = {h.name: h for h in self.heads}
        self.aux = []
        for key, src, flag, w, slot in (("cnotes", "notes", s.comp_notes, s.w_cnotes, S_CNOTES_LOSS),
                                        ("cinstr", "instr", s.comp_instr, s.w_cinstr, S_CINSTR_LOSS)):
            if flag:
                hsrc = self.head[src]
                a = _Aux(key, src, _Rec(key + ".rnn", hsrc.T, X_EXT, hsrc.N), w, slot)
                a.head.N = s.C
                self.aux.append(a)
        self.dec_heads = list(self.heads)          # the decoder's own heads (self.heads also lists the classifiers' Dense heads)
        self.heads = self.heads + [a.head for a in self.aux]
        enc_recs = [r for layer in self.enc_bi for r in layer] if self.enc_bi else self.enc_notes
        self.all_rec = (enc_recs + [m[0] for m in self.enc_meta] + [r for h in self.dec_heads for r in h.layers] +
                        [a.rec for a in self.aux])
        self.ncat = s.ncat
        self.has_pack = s.has_pack

    # ------------------------------------------------------------------------------------------------------
    # buffers (sized for max_batch; smaller batches reinterpret the same storage with a smaller row stride)
    # ------------------------------------------------------------------------------------------------------
    def _alloc_common(self, B):
        """buffers of the recurrent layers (self.all_rec) and of the output heads (self.heads), sized for B rows"""
        s, dev, dt = self.spec, self.device, self.dt
        H, GH = s.H, s.GH
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        st = self.store = {}

        def buf(name, n, **kw):
            st[name] = torch.zeros(int(n), **kw)

        esz = dict(dtype=dt, device=dev)
        # time-pipelined stacks: progress counters / ready flags (4 stack slots x 1024 words) and the time-out status word
        # slots 0-3: the pipelined stacks; 4: a single-layer branch followed by the K-streaming launch; 5: the expansion of a 1-feature roll
        st["sync"] = torch.zeros(16 * 1024, dtype=torch.int32, device=dev)     # (6.. : single-layer problems of a phase launch; 11..: single-layer branches whose gradients go in time portions)
        words = torch.zeros(2, dtype=torch.int32, device=dev)     # [live status of the running step, latched status since the last check]
        st["join_words"] = torch.zeros(8, dtype=torch.int32, device=dev)       # (Engine._join, MVAE_VALUE_JOIN)
        # z' of a fused history pre-pass lands here (a FIXED address: the step is then a replayable plan) and is copied to the
        # caller's rows afterwards (Engine.train_step_begin hist_fused)
        st["hist_zout"] = torch.zeros(B * getattr(self.spec, "Z", 0) if self.training else 0, dtype=torch.float32, device=dev)
        st["pipe_words"], st["pipe_status"], st["pipe_latched"] = words, words[0:1], words[1:2]
        for r in self.all_rec:
            p = r.prefix
            buf(p + ".u_pack", GH * H, **esz)
            buf(p + ".ut_pack", GH * H, **esz)
            buf(p + ".sh", B * H, **f32)                 # carried f32 state between time chunks (h, c, dh, dc)
            buf(p + ".sc", B * H, **f32)
            buf(p + ".hs", (r.T + 1) * B * H, **esz)
            if s.cell == "LSTM" and self.training:       # (cell states are saved for the backward pass only)
                buf(p + ".cs", (r.T + 1) * B * H, **esz)
            if self.training:
                buf(p + ".acts", r.T * B * GH, **esz)
                buf(p + ".da", r.T * B * GH, **esz)
                if s.cell == "GRU":
                    buf(p + ".rh", r.T * B * H, **esz)
            if r.xmode == hl.X_INDEX:
                buf(p + ".table", r.K * GH, **esz)
                if self._paired_table(r):       # (what the slot-interleaved LSTM / GRU kernels gather: hl.TABLE_PAIRED)
                    buf(p + ".table_p", r.K * GH, **esz)
                if self._index_as_dense(r):
                    buf(p + ".xp", r.T * B * GH, **esz)
            elif r.xmode == X_GATHER2:
                buf(p + ".table", (r.K - s.attach) * GH, **esz)      # W[:D0] + b (pitch rows), W[D0:] (attached instrument rows)
                buf(p + ".table2", s.attach * GH, **esz)
                buf(p + ".xp", r.T * B * GH, **esz)
            elif self._scalar_as_dense(r):
                buf(p + ".xp", r.T * B * GH, **esz)
            elif r.xmode == hl.X_DENSE:
                buf(p + ".xp", r.T * B * GH, **esz)
                buf(p + ".wt", GH * H, **esz)            # W^T (GH,H): forward projection, k-contiguous
                if self.training:
                    buf(p + ".wc", H * GH, **esz)        # W (H,GH) in dtype: backward dx, k-contiguous
                    buf(p + ".dx", r.T * B * H, **esz)   # gradient w.r.t. the lower layer's h sequence
            elif r.xmode == hl.X_CONST:
                buf(p + ".xp0", B * GH, **esz)
                if self.training:
                    buf(p + ".dxp0", B * GH, **f32)
            elif r.xmode == X_EXT:
                buf(p + ".xp", r.T * B * GH, **esz)
        for li, layer in enumerate(getattr(self, "enc_bi", ())):
            if li == 0:
                continue
            T = layer[0].T
            buf("enc.bi.%d.cat" % li, T * B * 2 * H, **esz)              # [forward | backward] outputs of the layer below, time-aligned
            if len(layer) > 1:
                buf("enc.bi.%d.cat_rev" % li, T * B * 2 * H, **esz)      # ... reversed in time: what this layer's backward RNN reads
            for r in layer:
                buf(r.prefix + ".wt2", GH * 2 * H, **esz)                # W^T (GH, 2H)
                if self.training:
                    buf(r.prefix + ".wc2", 2 * H * GH, **esz)            # W (2H, GH) in the compute dtype
                    buf(r.prefix + ".g1", T * B * H, **esz)              # d(input)[:, :H] and [:, H:] of this record
                    buf(r.prefix + ".g2", T * B * H, **esz)
            if self.training:
                buf("enc.bi.%d.dext_f" % li, T * B * H, **esz)           # gradient arriving at the forward / backward layer below
                buf("enc.bi.%d.dext_r" % li, T * B * H, **esz)
        for a in getattr(self, "aux", ()):
            if self.training:
                buf(a.key + ".dp", a.rec.T * B * a.rec.K, **f32)    # gradient w.r.t. the source head's probabilities
                buf(a.key + ".dh", B * H, **f32)                    # ... w.r.t. the classifier RNN's last state
        # heads
        for h in self.heads:
            h.NP = ops.head_np(h.N)
            n = h.name
            buf(n + ".wt", h.NP * H, **esz)
            buf(n + ".argmax", h.T * B, **u8)              # (the velocity head: round(p))
            if self.training:
                buf(n + ".dl", h.T * B * h.NP, **esz)
                buf(n + ".dhs", h.T * B * H, **esz)
                buf(n + ".wc", H * h.NP, **esz)            # W (H, NP): the head kernel's fused input gradient
            buf("out.%s_p" % n, h.T * B * h.N, **f32)      # inference outputs on request
        return buf

    def _alloc(self, B):
        s, dev, dt = self.spec, self.device, self.dt
        H, GH, Z, T, V = s.H, s.GH, s.Z, s.T, s.V
        f32 = dict(dtype=torch.float32, device=dev)
        buf = self._alloc_common(B)
        st = self.store
        self.np_notes = self.head["notes"].NP
        # encoder tail / latent / decoder initial states (all f32, (B, .) row-major)
        for name, n in (("cat", self.ncat * H), ("pack", H), ("extra", H), ("mu", Z), ("lv", Z), ("zh", s.zin),
                        ("style_p", max(s.C, 1)), ("S", self.n_init * H)):
            buf(name, B * n, **f32)
        if self.training:
            for name, n in (("dS", self.n_init * H), ("dzh", s.zin), ("dmu", Z), ("dlv", Z), ("dtail", H),
                            ("dtail2", H), ("dcat", self.ncat * H)):
                buf(name, B * n, **f32)
            # transposed f32 copies of the Dense kernels around the latent (fused backward chain, csrc/latent.hip)
            h1w = H // 2 if s.split else H
            buf("lat.wt_init", self.n_init * H * s.zin, **f32)
            buf("lat.wt_mu", Z * h1w, **f32)
            buf("lat.wt_lv", Z * (H - h1w if s.split else H), **f32)
            if s.extra_layer:
                buf("lat.wt_extra", H * s.tail_in, **f32)
            if self.has_pack:
                buf("lat.wt_pack", H * self.ncat * H, **f32)
        # inputs: ONE contiguous block (staging.Stager uploads it with a single copy from a pinned mirror); the "in.*"
        # buffers are typed views of it.  Regions are sized for max_batch; smaller batches use a prefix of each region.
        regions = [("in.x_idx", T * B, torch.uint8), ("in.y_idx", T * B, torch.uint8), ("in.i_idx", V * B, torch.uint8),
                   ("in.c_idx", B, torch.uint8), ("in.vel", T * B, torch.float32), ("in.eps", B * Z, torch.float32),
                   ("in.rw_notes", T * B, torch.float32), ("in.rw_instr", V * B, torch.float32),
                   ("in.rw_vel", T * B, torch.float32), ("in.rw_style", B, torch.float32),
                   ("in.start_notes", B * s.Dout, torch.float32), ("in.start_instr", B * s.ID, torch.float32),
                   ("in.start_vel", B, torch.float32), ("in.hist", B * Z, torch.float32), ("in.z", B * Z, torch.float32),
                   ("in.eps2", B * Z, torch.float32)]
        if s.meta_held:
            regions += [("in.d_idx", T * B, torch.uint8), ("in.rw_held", T * B, torch.float32), ("in.start_held", B * 2, torch.float32)]
        if s.meta_next:
            regions += [("in.n_idx", T * B, torch.uint8), ("in.rw_next", T * B, torch.float32),
                        ("in.start_next", B * s.Dout, torch.float32)]
        if self.enc_bi:
            regions += [("in.x_idx_rev", T * B, torch.uint8)]
        if s.attach:         # the second hot column of every two-hot row: input (within its block) and target (absolute column)
            regions += [("in.xa_idx", T * B, torch.uint8), ("in.ya_idx", T * B, torch.uint8)]
        if s.add_dim:
            regions += [("in.add", B * s.add_dim, torch.float32)]
        if s.signature:
            regions += [("in.sig", B * s.SD, torch.float32), ("in.rw_sig", B, torch.float32)]
            buf("sig.out", B * s.SD, **f32)
        for a in self.aux:
            regions += [("in.rw_" + a.key, B, torch.float32)]
        # the targets / row weights only the decoder heads read go LAST: staging can upload them in a second copy, converted on
        # the host while the encoder recurrences already run (Stager.stage(defer_targets=True))
        late = ("in.y_idx", "in.ya_idx", "in.n_idx", "in.rw_notes", "in.rw_instr", "in.rw_vel", "in.rw_held", "in.rw_next")
        regions = [r for r in regions if r[0] not in late] + [r for r in regions if r[0] in late]
        self._in_regions, off = {}, 0
        self._in_late_off = None
        for name, n, tdt in regions:
            if name in late and self._in_late_off is None:
                self._in_late_off = off
            nbytes = int(n) * (1 if tdt == torch.uint8 else 4)
            self._in_regions[name] = (off, nbytes, tdt)
            off += (nbytes + 255) // 256 * 256
        self._in_block = torch.zeros(off, dtype=torch.uint8, device=dev)
        for name, (o, nbytes, tdt) in self._in_regions.items():
            st[name] = self._in_block[o:o + nbytes].view(tdt)
        self._stager = None

