"""``VAE`` - the reference's model object, backed by the MI355X engine.

Keeps the surface the reference's scripts use (reference vae_definition.py:39-441, callers vae_training.py:47-130,
286-300,788-809,966-978 and vae_evaluation.py:2180-2199,2471-2483):

    model = VAE(); model.create(**61 keyword arguments)
    model.encoder.predict(x_list, batch_size=..., verbose=...)            -> z (n, latent)
    model.decoder.predict(x_list, batch_size=...)                         -> [notes, instrument, velocity] (or bare array)
    model.autoencoder.fit(x_list, y_list, epochs=1, batch_size=..., shuffle=False, sample_weight=..., verbose=...)
                                                                          -> object with .history (Keras key names)
    model.autoencoder.evaluate(x_list, y_list, batch_size=..., verbose=...) -> list aligned with .metrics_names
    model.autoencoder.predict(x_list, batch_size=...)                     -> [notes, instrument, velocity, style]
    .save_weights(path) / .load_weights(path, by_name=False) / .reset_states() / .summary()

Inputs are the host NumPy lists built by the packers (packers.py); they are never mutated.  The engine is created on
first use, sized for the largest batch seen.  Differences from Keras, stated plainly: weights are stored in this
repo's own container (``.npz`` content under whatever file name the caller passes - the reference's ``*.pickle``
files are Keras-HDF5, which cannot be read here); optimizer state is not saved (same as the reference).
"""
from __future__ import annotations

import json
from collections import OrderedDict

import numpy as np

from .layout import ModelSpec, ParamLayout, init_params, spec_from_create_kwargs


class History(object):
    def __init__(self):
        self.history = OrderedDict()
        self.epoch = []


def _is_onehot(a):
    a = np.asarray(a)
    return bool(np.all((a == 0) | (a == 1)) and np.all(a.sum(-1) == 1))


def _to_index(a, what):
    """(..., K) one-hot float array -> uint8 indices; anything else is outside the implemented input format."""
    a = np.asarray(a)
    if a.shape[-1] > 255 or not _is_onehot(a):
        raise NotImplementedError("%s must be one-hot rows of width <= 255 (reference layout, import_midi.py:255-262); "
                                  "dense / multi-hot rows need the dense input projection, which is not built" % what)
    return np.argmax(a, axis=-1).astype(np.uint8)


class _Shared(object):
    """State shared by the three model views: spec, parameters, engine."""

    def __init__(self, spec: ModelSpec, dtype, seed, device):
        self.spec, self.dtype, self.seed, self.device = spec, dtype, seed, device
        self.layout = ParamLayout.build(spec)
        self.params_host = init_params(spec, seed)     # authoritative copy until an engine exists
        self.engine = None
        self.rng = np.random.default_rng(seed + 1)

    def get_engine(self, batch, training=True):
        from .engine import Engine                      # imported lazily: needs the HIP library and a GPU
        need = max(int(batch), 16)
        if self.engine is None or self.engine.maxB < need or (training and not self.engine.training):
            params = self.engine.get_params() if self.engine is not None else self.params_host
            self.engine = Engine(self.spec, max_batch=need, dtype=self.dtype, device=self.device, seed=self.seed,
                                 training=True)
            self.engine.set_params(params)
        return self.engine

    def current_params(self):
        return self.engine.get_params() if self.engine is not None else self.params_host

    def set_params(self, named):
        self.params_host = OrderedDict((k, np.asarray(v, np.float32)) for k, v in named.items())
        if self.engine is not None:
            self.engine.set_params(self.params_host)

    def epsilon(self, n):
        return (self.rng.standard_normal((n, self.spec.Z)) * self.spec.epsilon_std).astype(np.float32)


class _ModelView(object):
    """Common Keras-Model methods (weights, states, summary)."""
    prefixes = ("enc.", "dec.")
    name = "model"

    def __init__(self, shared: _Shared):
        self._s = shared

    def _names(self):
        return [n for n in self._s.layout.oracle_names() if n.startswith(self.prefixes)]

    def get_weights(self):
        p = self._s.current_params()
        return [p[n] for n in self._names()]

    def save_weights(self, filepath, overwrite=True):
        p = self._s.current_params()
        arrays = {n: p[n] for n in self._names()}
        arrays["__spec__"] = np.frombuffer(json.dumps(self._s.spec.__dict__, sort_keys=True).encode(), dtype=np.uint8)
        with open(filepath, "wb") as f:
            np.savez(f, **arrays)

    def load_weights(self, filepath, by_name=False):
        with np.load(filepath) as z:
            have = {k: z[k] for k in z.files if k != "__spec__"}
        p = OrderedDict(self._s.current_params())
        for n in self._names():
            if n not in have:
                raise ValueError("%s: tensor %r missing in %s" % (self.name, n, filepath))
            if have[n].shape != p[n].shape:
                raise ValueError("%s: shape of %r is %s in the file, %s in the model" % (self.name, n, have[n].shape, p[n].shape))
            p[n] = have[n]
        self._s.set_params(p)

    def reset_states(self):
        """No layer of this model is stateful (reference vae_training.py:811-812 calls it regardless)."""

    def count_params(self):
        return int(sum(int(np.prod(self._s.layout.entries[n].shape)) for n in self._names()))

    def summary(self):
        lines = ["Model: %s  (%s cells, H=%d, Z=%d, T=%d)" % (self.name, self._s.spec.cell, self._s.spec.H, self._s.spec.Z,
                                                            self._s.spec.T),
                 "%-32s %-18s %10s" % ("tensor", "shape", "params")]
        for n in self._names():
            shp = self._s.layout.entries[n].shape
            lines.append("%-32s %-18s %10d" % (n, shp, int(np.prod(shp))))
        lines.append("Total params: %d" % self.count_params())
        return "\n".join(lines)


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


class Encoder(_ModelView):
    prefixes = ("enc.",)
    name = "encoder"

    def _unpack(self, x):
        sp = self._s.spec
        x = _as_list(x)
        i = 0
        X = x[i]; i += 1
        I = x[i] if sp.meta_instrument else None
        i += int(sp.meta_instrument)
        V = x[i] if sp.meta_velocity else None
        return (_to_index(X, "notes input"), _to_index(I, "instrument input") if I is not None else None,
                np.asarray(V, np.float32)[..., 0] if V is not None else None)

    def predict(self, x, batch_size=32, verbose=0):
        """Sampled z for every window (fresh epsilon per call, reference vae_definition.py:498-502)."""
        x_idx, i_idx, vel = self._unpack(x)
        n = x_idx.shape[0]
        out = np.zeros((n, self._s.spec.Z), np.float32)
        eng = self._s.get_engine(min(batch_size, max(n, 1)))
        for lo in range(0, n, batch_size):
            hi = min(n, lo + batch_size)
            B = eng.stage_encoder_inputs(x_idx[lo:hi], None if i_idx is None else i_idx[lo:hi],
                                         None if vel is None else vel[lo:hi], self._s.epsilon(hi - lo))
            out[lo:hi] = eng.encode(B).cpu().numpy()
        return out


class Decoder(_ModelView):
    prefixes = ("dec.",)
    name = "decoder"

    def _unpack(self, x):
        """[start, z, (history), instr_start, vel_start]  (reference vae_definition.py:820-865)"""
        sp = self._s.spec
        x = _as_list(x)
        start, z = x[0], x[1]
        i = 2
        hist = x[i] if sp.history else None
        i += int(sp.history)
        istart = x[i] if sp.meta_instrument else None
        i += int(sp.meta_instrument)
        vstart = x[i] if sp.meta_velocity else None
        return start, z, hist, istart, vstart

    def _run(self, x, batch_size, want_probs=True):
        sp = self._s.spec
        start, z, hist, istart, vstart = self._unpack(x)
        n = np.asarray(z).shape[0]
        eng = self._s.get_engine(min(batch_size, max(n, 1)))
        outs, idxs = [], []
        for lo in range(0, n, batch_size):
            hi = min(n, lo + batch_size)
            B = hi - lo
            eng.stage_decoder_inputs(B, hist=None if hist is None else np.asarray(hist)[lo:hi], z=np.asarray(z)[lo:hi],
                                     start_notes=np.asarray(start)[lo:hi],
                                     start_instr=None if istart is None else np.asarray(istart)[lo:hi],
                                     start_vel=None if vstart is None else np.asarray(vstart)[lo:hi])
            eng.decode(B, want_probs=want_probs)
            if want_probs:
                outs.append(eng.outputs(B))
            idxs.append(eng.note_indices(B))
        return outs, np.concatenate(idxs, 0) if idxs else np.zeros((0, sp.T), np.uint8)

    def predict(self, x, batch_size=32, verbose=0):
        sp = self._s.spec
        outs, _ = self._run(x, batch_size)
        res = [np.concatenate([o["notes"] for o in outs], 0)]
        if sp.meta_instrument:
            res.append(np.concatenate([o["instr"] for o in outs], 0))
        if sp.meta_velocity:
            res.append(np.concatenate([o["vel"] for o in outs], 0))
        return res if len(res) > 1 else res[0]

    def predict_note_indices(self, x, batch_size=32):
        """Fused argmax decode: (n, T) uint8 note index per row, computed on the device without materialising the
        (n, T, D) probability tensor on the host (replaces decoder.predict + sample_vector 'argmax')."""
        return self._run(x, batch_size, want_probs=False)[1]


class Autoencoder(_ModelView):
    name = "autoencoder"

    def __init__(self, shared, encoder: Encoder):
        super().__init__(shared)
        self._enc = encoder

    # ---- names -------------------------------------------------------------------------------------------
    def _decoder_outputs(self):
        sp = self._s.spec
        return ["notes"] + (["instr"] if sp.meta_instrument else []) + (["vel"] if sp.meta_velocity else [])

    @property
    def metrics_names(self):
        """Keras 2.0.8 naming: repeated 'decoder_loss' / 'decoder_acc' per decoder output; the reference
        de-duplicates them itself (vae_training.py:172-187)."""
        sp = self._s.spec
        k = len(self._decoder_outputs())
        single = k == 1 and not sp.style
        if single:
            return ["loss", "acc"]
        names = ["loss"] + ["decoder_loss"] * k + (["composer_decoder_loss"] if sp.style else [])
        names += ["decoder_acc"] * k + (["composer_decoder_acc"] if sp.style else [])
        return names

    def _history_keys(self):
        sp = self._s.spec
        outs = self._decoder_outputs()
        if len(outs) == 1 and not sp.style:
            return [("loss", "loss"), ("acc", "notes_acc")]
        keys = [("loss", "loss")]
        if len(outs) == 1:
            keys += [("decoder_loss", "notes_loss"), ("decoder_acc", "notes_acc")]
        else:
            for i, o in enumerate(outs, 1):
                keys += [("decoder_loss_%d" % i, o + "_loss"), ("decoder_acc_%d" % i, o + "_acc")]
        if sp.style:
            keys += [("composer_decoder_loss", "style_loss"), ("composer_decoder_acc", "style_acc")]
        return keys

    # ---- list unpacking (orders of reference vae_definition.py:924-1040) -------------------------------------
    def _unpack_x(self, x):
        sp = self._s.spec
        x = _as_list(x)
        X, start = x[0], x[1]
        i = 2
        hist = x[i] if sp.history else None
        i += int(sp.history)
        istart = I = vstart = V = None
        if sp.meta_instrument:
            istart, I = x[i], x[i + 1]
            i += 2
        if sp.meta_velocity:
            vstart, V = x[i], x[i + 1]
        return X, start, hist, istart, I, vstart, V

    def _unpack_y(self, y):
        sp = self._s.spec
        y = _as_list(y)
        Y = y[0]
        i = 1 + int(sp.meta_instrument) + int(sp.meta_velocity)
        C = y[i] if sp.style else None
        return Y, C

    def _unpack_w(self, w, n):
        """[w_notes (n,T), w_style, w_instr, w_vel] - the reference's order (vae_definition.py:930-1004)."""
        sp = self._s.spec
        if w is None:
            return None, None, None, None
        w = _as_list(w) if isinstance(w, (list, tuple)) else [w]
        wn = w[0]
        i = 1
        ws = w[i] if sp.style and len(w) > i else None
        i += int(sp.style)
        wi = w[i] if sp.meta_instrument and len(w) > i else None
        i += int(sp.meta_instrument)
        wv = w[i] if sp.meta_velocity and len(w) > i else None
        return wn, ws, wi, wv

    def _stage(self, eng, lo, hi, xs, ys, ws):
        X_idx, start, hist, istart, I_idx, vstart, vel, Y_idx, C_idx = xs + ys
        sl = slice(lo, hi)
        B = eng.stage_encoder_inputs(X_idx[sl], None if I_idx is None else I_idx[sl], None if vel is None else vel[sl],
                                     self._s.epsilon(hi - lo))
        eng.stage_decoder_inputs(B, hist=None if hist is None else np.asarray(hist)[sl], start_notes=np.asarray(start)[sl],
                                 start_instr=None if istart is None else np.asarray(istart)[sl],
                                 start_vel=None if vstart is None else np.asarray(vstart)[sl])
        if Y_idx is not None:
            wn, wst, wi, wv = ws
            eng.stage_targets(B, Y_idx[sl], None if C_idx is None else C_idx[sl],
                              w_notes=None if wn is None else np.asarray(wn)[sl],
                              w_instr=None if wi is None else np.asarray(wi)[sl],
                              w_vel=None if wv is None else np.asarray(wv)[sl],
                              w_style=None if wst is None else np.asarray(wst)[sl])
        return B

    def _prepare(self, x, y):
        X, start, hist, istart, I, vstart, V = self._unpack_x(x)
        xs = (_to_index(X, "notes input"), start, hist, istart, _to_index(I, "instrument input") if I is not None else None,
              vstart, np.asarray(V, np.float32)[..., 0] if V is not None else None)
        if y is None:
            return xs, (None, None)
        Y, C = self._unpack_y(y)
        return xs, (_to_index(Y, "notes target"), _to_index(C, "style target") if C is not None else None)

    # ---- Keras methods -----------------------------------------------------------------------------------------
    def fit(self, x, y, epochs=1, batch_size=32, shuffle=False, sample_weight=None, verbose=0, allreduce=None):
        """One optimizer step per minibatch of consecutive windows (reference vae_training.py:804-809).  History
        values are batch-size-weighted means over the minibatches of each epoch (Keras BaseLogger semantics)."""
        if shuffle:
            raise NotImplementedError("shuffle=True (the reference always passes shuffle=False)")
        xs, ys = self._prepare(x, y)
        n = xs[0].shape[0]
        ws = self._unpack_w(sample_weight, n)
        eng = self._s.get_engine(min(batch_size, max(n, 1)), training=True)
        hist = History()
        keys = self._history_keys()
        for e in range(epochs):
            tot = OrderedDict((k, 0.0) for k, _ in keys)
            for lo in range(0, n, batch_size):
                hi = min(n, lo + batch_size)
                B = self._stage(eng, lo, hi, xs, ys, ws)
                eng.train_step(B, allreduce=allreduce)
                m = eng.metrics(B)
                for k, src in keys:
                    tot[k] += m[src] * B
            for k in tot:
                hist.history.setdefault(k, []).append(tot[k] / max(n, 1))
            hist.epoch.append(e)
        return hist

    def _forward_all(self, x, y, batch_size, want_probs):
        xs, ys = self._prepare(x, y)
        n = xs[0].shape[0]
        eng = self._s.get_engine(min(batch_size, max(n, 1)))
        keys = self._history_keys()
        tot = OrderedDict((k, 0.0) for k, _ in keys)
        outs = []
        for lo in range(0, n, batch_size):
            hi = min(n, lo + batch_size)
            B = self._stage(eng, lo, hi, xs, ys, (None, None, None, None))
            if y is None:
                eng._have_targets = False
                eng.scal.zero_()
                if eng._weights_dirty:
                    eng.prepare_weights()
                eng.encoder_forward(B)
                eng.decoder_forward(B, want_probs=True)
            else:
                eng.eval_step(B, want_probs=want_probs)
                m = eng.metrics(B)
                for k, src in keys:
                    tot[k] += m[src] * B
            if want_probs:
                outs.append(eng.outputs(B))
        return tot, outs, n

    def evaluate(self, x, y, batch_size=32, verbose=0, sample_weight=None):
        tot, _, n = self._forward_all(x, y, batch_size, want_probs=False)
        by_key = {k: v / max(n, 1) for k, v in tot.items()}
        keys = [k for k, _ in self._history_keys()]
        losses = [k for k in keys if "loss" in k]
        accs = [k for k in keys if "acc" in k]
        return [by_key[k] for k in losses + accs]

    def predict(self, x, batch_size=32, verbose=0):
        sp = self._s.spec
        _, outs, _ = self._forward_all(x, None, batch_size, want_probs=True)
        res = [np.concatenate([o["notes"] for o in outs], 0)]
        if sp.meta_instrument:
            res.append(np.concatenate([o["instr"] for o in outs], 0))
        if sp.meta_velocity:
            res.append(np.concatenate([o["vel"] for o in outs], 0))
        if sp.style:
            res.append(np.concatenate([o["style"] for o in outs], 0))
        return res if len(res) > 1 else res[0]


class VAE(object):
    """reference vae_definition.py:39.  ``create`` takes the reference's keyword arguments unchanged; two extra,
    optional ones select the arithmetic (``compute_dtype`` 'bf16' | 'f32') and the initialiser seed."""

    def __init__(self):
        self.encoder = self.decoder = self.autoencoder = self.composer_decoder = None

    def create(self, compute_dtype="bf16", seed=0, device="cuda:0", **kw):
        for k, v in kw.items():
            setattr(self, k, v)
        self.spec = spec_from_create_kwargs(kw)
        shared = _Shared(self.spec, compute_dtype, seed, device)
        self._shared = shared
        self.encoder = Encoder(shared)
        self.decoder = Decoder(shared)
        self.autoencoder = Autoencoder(shared, self.encoder)
        self.composer_decoder = None     # parameter-free softmax over z[:, :C]; reported through the autoencoder
        return self
