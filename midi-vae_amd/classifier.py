"""The reference's three style classifiers on the MI355X engine.

reference pitch_classifier.py:89-103, velocity_classifier.py:110-125, instrument_classifier.py:93-107: the same model three
times - ``num_layers`` Keras GRU(lstm_size) layers over a roll (one-hot pitch rows, the velocity roll, one-hot instrument rows)
and Dense(num_classes, softmax) on the last hidden state; categorical cross-entropy, accuracy, Keras Adam; trained one song
per ``fit`` call (pitch_classifier.py:223-245), scored with evaluate / predict and a confusion matrix (:117-160).  They are what
turns the decoder's style-transfer output into a score.

``ClassifierEngine`` is the VAE engine's machinery on this smaller graph: the resident-weights recurrent kernels and
time-pipelined stacks (csrc/rnn_resident.hip), the fused softmax / loss / accuracy / argmax head kernel on the last state, the
one-hot-table and weight-gradient GEMMs, the Keras-Adam kernel.  ``StyleClassifier`` keeps the Keras ``Model`` calls the
reference scripts make: fit / evaluate / predict / reset_states / save / summary.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import hiplib as hl
from . import ops
from .engine import Engine, _Head, _Rec
from .layout import ClassifierSpec, classifier_layout, init_classifier_params
from .model import History
from .staging import host_onehot_to_index

S_LOSS, S_HITS = 0, 1


class ClassifierEngine(Engine):
    def _make_layout(self):
        return classifier_layout(self.spec)

    def _initial_params(self, seed):
        return init_classifier_params(self.spec, seed)

    def _build_graph_description(self):
        s = self.spec
        first = hl.X_INDEX if s.xmode == "index" else hl.X_SCALAR
        self.layers = []
        for l in range(s.L):
            self.layers.append(_Rec("rnn.%d" % l, s.T, first if l == 0 else hl.X_DENSE, s.K if l == 0 else s.H,
                                    lower=self.layers[-1] if l else None))
        cls = _Head("cls", [self.layers[-1]], 0, s.C, 1.0, S_LOSS, "in.c_idx")
        cls.T, cls.out = 1, "cls.out"                  # ONE row per window: the Dense reads the last hidden state only
        self.heads, self.head = [cls], {"cls": cls}
        self.all_rec = list(self.layers)
        self.enc_notes = self.layers                   # (what the shared stream-count / pipelining helpers look at)

    def _alloc(self, B):
        s, dev = self.spec, self.device
        buf = self._alloc_common(B)
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        if s.xmode == "index":
            buf("in.x_idx", s.T * B, **u8)
        else:
            buf("in.x_val", s.T * B, **f32)
        buf("in.c_idx", B, **u8)
        buf("in.rw", B, **f32)
        buf("h_last", B * s.H, **f32)                  # final hidden state, f32 (the recurrent kernels' h_last output)
        buf("h_top", B * s.H, dtype=self.dt, device=dev)    # ... in the compute dtype: what the head kernel reads
        if self.training:
            buf("dh_last", B * s.H, **f32)

    # ---- staging -----------------------------------------------------------------------------------------------
    def stage(self, x, c_idx=None, norm_B=None):
        """x: (B,T) uint8 indices or (B,T) float values; c_idx (B,) uint8 class per window (None: no targets)"""
        s = self.spec
        B = x.shape[0]
        Bp = self.pad16(B)
        if s.xmode == "index":
            self._up_tm("in.x_idx", np.asarray(x, np.uint8), torch.uint8)
        else:
            self._up_tm("in.x_val", np.asarray(x, np.float32), torch.float32)
        self._have_targets = c_idx is not None
        self.norm_B = float(B if norm_B is None else norm_B)
        if c_idx is not None:
            c = np.full((Bp,), 255, np.uint8)
            c[:B] = c_idx
            self._up("in.c_idx", c, torch.uint8)
            rw = np.zeros((Bp,), np.float32)
            rw[:B] = 1.0 / self.norm_B
            self._up("in.rw", rw, torch.float32)
        return B

    # ---- forward / backward ------------------------------------------------------------------------------------
    def forward(self, B, want_probs=False, verify=None):
        """``verify``: check the first time-pipelined use of a forward-only call (evaluate / predict) as Engine.encode does; a
        train step verifies once, after its backward pass"""
        s, P = self.spec, self.P
        Breal0 = B
        Breal, B = B, self.pad16(B)
        if self._weights_dirty:
            self.prepare_weights()
        else:
            self.scal.zero_()
        inp = (dict(idx=self._v("in.x_idx", s.T, B)) if s.xmode == "index" else dict(xs=self._v("in.x_val", s.T, B)))
        self._stack_forward(self.layers, B, h_last=self._v("h_last", B, s.H), h_last_ld=s.H, slot=0, **inp)
        self._prefork = None
        top = self._v(self.layers[-1].prefix + ".hs", s.T + 1, B, s.H)[s.T]      # (B,H) in the compute dtype
        h = self.head["cls"]
        tg = self._have_targets
        ops.head(0, self.kind, B, s.H, s.C, top, self._v("cls.wt", h.NP, s.H), P["cls.out.b"],
                 target_idx=self._v("in.c_idx", B) if tg else None, row_weight=self._v("in.rw", B) if tg else None,
                 grad_scale=1.0, probs=self._v("out.cls_p", B, s.C) if want_probs else None, argmax=self._v("cls.argmax", B),
                 dlogits=self._v("cls.dl", B, h.NP) if (self.training and tg) else None, scalars=self.scal[S_LOSS:S_LOSS + 2],
                 b_stride=B, b_valid=Breal)
        if verify:
            self._verify_pipeline(lambda: (self.scal.zero_(), self.forward(Breal0, want_probs, verify=False)), key="cls_fwd")

    def backward(self, B):
        s, P, G = self.spec, self.P, self.G
        B = self.pad16(B)
        h = self.head["cls"]
        dl = self._v("cls.dl", B, h.NP)
        top = self._v(self.layers[-1].prefix + ".hs", s.T + 1, B, s.H)[s.T]
        dh = self._v("dh_last", B, s.H)
        ops.gemm(dl, self._v("cls.wt", h.NP, s.H), dh, B, s.H, h.NP)                 # d(last hidden state), f32
        self._fork(self.s_grad)
        with self._on(self.s_grad):
            ops.gemm(top, dl, G["cls.out.W"], s.H, s.C, B, trans_a=True, ldb=h.NP, accumulate=True)
            ops.colsum(dl, B, s.C, G["cls.out.b"], ldx=h.NP)
        inp = (dict(idx=self._v("in.x_idx", s.T, B)) if s.xmode == "index" else dict(xs=self._v("in.x_val", s.T, B)))
        self._stack_backward(self.layers, B, dh_last=dh, dh_last_ld=s.H, slot=3, **inp)
        self._prefork = None
        self._join(self.s_grad)
        self._join(self.s_grad2)

    def train_step(self, B, allreduce=None):
        assert self.training
        if not self._grads_clean:
            self.grads.zero_()
        self._grads_clean = False
        self.forward(B)
        self.backward(B)
        self._verify_pipeline(lambda: (self.grads.zero_(), self.scal.zero_(), self.forward(B, verify=False), self.backward(B)),
                              key="cls_train")
        gs = allreduce(self.grads) if allreduce is not None else 1.0
        self.optimizer_step(gs if gs is not None else 1.0)

    # ---- results -------------------------------------------------------------------------------------------------
    HIT_MASK = 1 << S_HITS

    def _metrics_from(self, v, B):
        return OrderedDict(loss=v[S_LOSS], acc=v[S_HITS] / B)

    def probs(self, B):
        return self._v("out.cls_p", self.pad16(B), self.spec.C)[:B].cpu().numpy()


class StyleClassifier(object):
    """Keras ``Model`` surface of the reference's classifier scripts.  ``kind``: 'pitch' (one-hot (n,T,61) rolls), 'velocity'
    ((n,T,1) velocity rolls), 'instrument' ((n,V,16) one-hot rows)."""

    def __init__(self, kind="pitch", input_dim=61, num_classes=2, lstm_size=256, num_layers=2, learning_rate=2e-5, optimizer="Adam",
                 compute_dtype="bf16", seed=0, device="cuda:0"):
        if kind not in ("pitch", "velocity", "instrument"):
            raise ValueError(kind)
        self.kind = kind
        self.xmode = "scalar" if kind == "velocity" else "index"
        self.cfg = dict(K=1 if kind == "velocity" else int(input_dim), C=int(num_classes), H=int(lstm_size), L=int(num_layers),
                        lr=float(learning_rate), optimizer={"RMS": "RMSprop"}.get(optimizer, optimizer), xmode=self.xmode)
        self.dtype, self.seed, self.device = compute_dtype, seed, device
        self.engine = None
        self._params = None
        self.metrics_names = ["loss", "acc"]

    def _engine(self, T, batch):
        need = max(int(batch), 16)
        if self.engine is None or self.engine.spec.T != T or self.engine.maxB < need:
            old = self.engine
            params = old.get_params() if old is not None else self._params
            opt = old.get_optimizer_state() if old is not None else None
            self.engine = None
            del old
            spec = ClassifierSpec(T=int(T), **self.cfg)
            self.engine = ClassifierEngine(spec, max_batch=need, dtype=self.dtype, device=self.device, seed=self.seed)
            if params is not None:
                self.engine.set_params(params)
            if opt is not None:
                self.engine.set_optimizer_state(opt)
        return self.engine

    def _inputs(self, X):
        X = np.asarray(X)
        if X.ndim == 2 and X.dtype == np.uint8 and self.xmode == "index":
            return X                                   # already note / instrument indices (e.g. the decoder's fused argmax)
        if X.ndim != 3:
            raise ValueError("expected (n, T, input_dim) rolls, got %s" % (X.shape,))
        if self.xmode == "scalar":
            return X[..., 0].astype(np.float32)
        if X.shape[2] != self.cfg["K"]:
            raise ValueError("input_dim is %d, got rows of width %d" % (self.cfg["K"], X.shape[2]))
        return host_onehot_to_index(X, "%s roll" % self.kind)

    @staticmethod
    def _targets(Y, n):
        Y = np.asarray(Y)
        if Y.ndim == 1:
            Y = np.tile(Y[None], (n, 1))           # (the instrument script passes one bare one-hot row per song)
        return host_onehot_to_index(Y[:, None, :].astype(np.float64), "class target")[:, 0]

    def fit(self, X, Y, epochs=1, batch_size=32, shuffle=False, verbose=0):
        """one optimizer step per minibatch of consecutive windows (reference pitch_classifier.py:234-238)"""
        if shuffle:
            raise NotImplementedError("shuffle=True (the reference passes shuffle=False)")
        x = self._inputs(X)
        n, T = x.shape
        c = self._targets(Y, n)
        eng = self._engine(T, batch_size)
        pending = []
        for e in range(epochs):
            pending.append(eng.reset_accumulated())
            for lo in range(0, n, batch_size):
                hi = min(n, lo + batch_size)
                B = eng.stage(x[lo:hi], c[lo:hi])
                eng.train_step(B)
                eng.accumulate_metrics(hi - lo)

        def resolve(out):
            for acc in pending:
                m = eng.read_accumulated(n, acc=acc)
                for k in ("loss", "acc"):
                    out.setdefault(k, []).append(m[k])

        h = History(resolve)
        h.epoch = list(range(epochs))
        return h

    def evaluate(self, X, Y, batch_size=32, verbose=0):
        x = self._inputs(X)
        n, T = x.shape
        c = self._targets(Y, n)
        eng = self._engine(T, batch_size)
        eng.reset_accumulated()
        for lo in range(0, n, batch_size):
            hi = min(n, lo + batch_size)
            B = eng.stage(x[lo:hi], c[lo:hi])
            eng.forward(B, verify=True)
            eng.accumulate_metrics(hi - lo)
        m = eng.read_accumulated(n)
        return [m["loss"], m["acc"]]

    def predict(self, X, batch_size=32, verbose=0):
        x = self._inputs(X)
        n, T = x.shape
        eng = self._engine(T, batch_size)
        out = []
        for lo in range(0, n, batch_size):
            hi = min(n, lo + batch_size)
            B = eng.stage(x[lo:hi], None)
            eng.forward(B, want_probs=True, verify=True)
            out.append(eng.probs(B))
        eng.check_pipeline()
        return np.concatenate(out, 0) if out else np.zeros((0, self.cfg["C"]), np.float32)

    def reset_states(self):
        """no stateful layers (the reference calls it regardless, pitch_classifier.py:240-241)"""

    def get_weights(self):
        p = self.engine.get_params() if self.engine is not None else self._params
        return None if p is None else list(p.values())

    def save(self, filepath):
        p = self.engine.get_params()
        with open(filepath, "wb") as f:
            np.savez(f, **p)

    def load_weights(self, filepath):
        with np.load(filepath) as z:
            self._params = OrderedDict((k, z[k]) for k in z.files)
        if self.engine is not None:
            self.engine.set_params(self._params)

    def summary(self):
        c = self.cfg
        return "StyleClassifier(%s): %d x GRU(%d) over (T, %d) -> Dense(%d, softmax)" % (self.kind, c["L"], c["H"], c["K"], c["C"])
