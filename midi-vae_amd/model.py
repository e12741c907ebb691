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

Inputs are the host NumPy lists built by the packers (packers.py); they are never mutated.  They are converted by the
multi-threaded host packers of the C ABI (staging.py / csrc/hostpack.cpp) straight into a pinned staging block, one
asynchronous upload per minibatch; losses and metrics accumulate on the device and are read back once per fit call.
The engine is created on first use, sized for the ``batch_size`` of the call.  Differences from Keras, stated plainly:
weights are stored in this repo's own container (``.npz`` content under whatever file name the caller passes - the
reference's ``*.pickle`` files are Keras-HDF5, which cannot be read here); optimizer state is not saved (same as the
reference).
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import numpy as np

from .layout import ModelSpec, ParamLayout, init_params, spec_from_create_kwargs


class History(object):
    """What ``fit`` returns.  ``history`` (Keras: key -> list of per-epoch values) is filled on FIRST ACCESS: the losses and metrics
    of a fit call are accumulated on the device, and reading them is the only point at which the host waits for the device.  The
    reference reads ``hist.history[...]`` right after every ``fit`` (vae_training.py:817-864) and gets Keras' behaviour; a caller
    that keeps the History objects of several songs and reads them later (this repo's vae_training.py: at the end of the epoch)
    lets the next song's host work overlap the device's work on this one."""

    def __init__(self, resolve=None):
        self._resolve, self._history = resolve, OrderedDict()
        self.epoch = []

    @property
    def history(self):
        if self._resolve is not None:
            resolve, self._resolve = self._resolve, None
            resolve(self._history)
        return self._history


class DeviceLatent(object):
    """Sampled z of every window of a song, resident in HBM - what ``encoder.predict(..., device=True)`` returns instead of a
    host array.  Passed in the history slot of the fit / evaluate input list it stands for the ROLLED history of reference
    vae_training.py:795-798 (H[1:] = z[:-1], H[0] = 0): the engine places row i-1 beside window i on the device, so the
    history pre-pass costs no device->host->device round trip.  ``numpy()`` gives the rolled host array the reference builds.

    The object is DEFERRED when it is created: the draw of the pre-pass (a fresh epsilon per window, taken from the model's
    generator at the time of the ``predict`` call, so the random stream is the reference's) and the caller's arrays are kept,
    the encoder has not run.  ``autoencoder.fit`` on the same song then takes the history of its FIRST minibatch out of that
    minibatch's own encoder forward - same weights, same windows, same mu / log sigma^2, two draws (reference
    vae_definition.py:498-502) - and runs the encoder once, at a chip-filling batch, over the windows of the later
    minibatches before its first update (model.Autoencoder.fit); a song of one minibatch pays no second encoder forward at
    all.  Any other use (evaluate, latent(), numpy(), a fit on other windows, data parallel runs) runs the pre-pass when it
    is first needed.  Either way the weights must be those of the ``predict`` call: a model updated in between raises."""

    def __init__(self, z_dev=None, shared=None, arrays=None, eps=None, parent=None, rows=None):
        self._parent = parent
        if parent is not None:                        # a prefix view (packers: H[:-1] with the next-notes head)
            self.shape = (rows, parent.shape[1])
            return
        self._shared, self._arrays, self.eps = shared, arrays, eps
        if z_dev is not None:
            self._z, self._valid = z_dev, True
        else:
            import torch
            self._z = torch.zeros(eps.shape, dtype=torch.float32, device=shared.engine_device())
            self._valid = False
            self._version = shared.param_version()
        self.shape = tuple(self._z.shape)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, sl):
        """prefix slices only (the packers drop the last window when the next-notes head is on: H[:-1])"""
        if not (isinstance(sl, slice) and sl.start in (None, 0) and sl.step in (None, 1)):
            raise IndexError("DeviceLatent supports prefix slices only; use .numpy() for anything else")
        rows = len(range(*sl.indices(self.shape[0])))
        return DeviceLatent(parent=self._root(), rows=rows)

    def _root(self):
        return self if self._parent is None else self._parent

    @property
    def deferred(self):
        return not self._root()._valid

    def check_version(self):
        r = self._root()
        if not r._valid and r._version != r._shared.param_version():
            raise RuntimeError("the model's weights changed between encoder.predict(device=True) and the use of its result: the "
                               "history pre-pass must see the weights fit starts from (reference vae_training.py:795-809)")

    @staticmethod
    def _same_array(A, A0, rows=None):
        """the same memory, shape (up to a row prefix), dtype and strides - or both absent"""
        if A is None or A0 is None:
            return A is None and A0 is None
        return (isinstance(A, np.ndarray) and isinstance(A0, np.ndarray) and A.shape[1:] == A0.shape[1:] and A.dtype == A0.dtype and
                A.__array_interface__["data"][0] == A0.__array_interface__["data"][0] and A.strides == A0.strides and
                (rows is None or A.shape[0] == rows))

    def matches(self, X, n, I=None, Vel=None, Held=None):
        """may a fit call on these arrays take its history out of its own encoder forward?  Only if they ARE the arrays this latent
        was requested for (the pre-pass of the later minibatches and the first minibatch's own forward must encode what
        encoder.predict was given: notes AND instrument / velocity / held rolls - ADVICE r03) and the view covers the WHOLE
        root: a prefix view (H[:-1] with the next-notes head) would leave the root's last row unencoded but marked valid."""
        r = self._root()
        if r._arrays is None or n != self.shape[0] or self.shape[0] != r.shape[0]:
            return False
        X0, I0, Vel0, Held0 = r._arrays

        def same(A, A0):
            # the reference's packers build the encoder list and the autoencoder list separately (prepare_encoder_input_list /
            # prepare_autoencoder_input_and_output_list, vae_definition.py:770-1045): X is passed through (same memory), the meta
            # rolls are fresh copies - compared by content (a few MB per song)
            if A is None or A0 is None:
                return A is None and A0 is None
            return self._same_array(A, A0) or (np.shape(A) == np.shape(A0) and np.array_equal(A, A0))
        return self._same_array(X, X0, n) and same(I, I0) and same(Vel, Vel0) and same(Held, Held0)

    def mark_valid(self):
        r = self._root()
        r._valid, r._arrays = True, None

    @property
    def z(self):
        """(n, Z) f32 device tensor, UNROLLED; runs the pre-pass if it has not run yet"""
        r = self._root()
        if not r._valid:
            r.check_version()
            X, I, Vel, Held = r._arrays
            r._shared.encode_windows(X, I, Vel, Held, r.eps, 0, r.shape[0], r._z)
            r.mark_valid()
        return r._z[:self.shape[0]]

    def latent(self):
        return self.z.cpu().numpy()

    def numpy(self):
        z = self.latent()
        H = np.zeros_like(z)
        H[1:] = z[:-1]
        return H


class EpsilonStream(object):
    """The model's N(0, epsilon_std) draws as float32 (n, Z) blocks in CALL order, generated AHEAD of the caller on a worker thread
    (reference: K.random_normal inside the sampling Lambda, vae_definition.py:498-502 - one draw per batch).  ONE stream: numpy fills
    an array sample by sample, so R rows drawn at once and handed out in order are the values R successive draws give - the oracle
    tests draw minibatch by minibatch from an equal generator and get equal numbers.  At the reference's default settings the two
    draws of a step (the step's own, the history pre-pass's) were 0.45 ms of a 1.7 ms step on the caller's thread.

    The stream runs AHEAD of its consumer by up to 1.5 blocks, so the generator's own state is not "what has been consumed".  A
    caller that touches the generator between two draws is noticed (its state no longer is what the last block left behind): the
    buffered rows are dropped and the next draw starts from the generator as it is now - an in-place reseed or a restored state
    takes effect at once, as it would with one draw per batch.  ``reset()`` does the same explicitly; ``close()`` ends the worker
    thread (a replaced stream used to leave its executor behind)."""
    BLOCK = 1024

    def __init__(self, rng, Z, std):
        from concurrent.futures import ThreadPoolExecutor
        self.rng, self.Z, self.std = rng, int(Z), float(std)
        self._buf, self._pos, self._next = np.zeros((0, self.Z), np.float32), 0, None
        self._left = self._fingerprint()       # the generator's state behind the last block drawn
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mvae-epsilon")      # (every draw, in submission order)

    def _fingerprint(self):
        return repr(self.rng.bit_generator.state)

    def _draw(self, rows):
        out = (self.rng.standard_normal((rows, self.Z)) * self.std).astype(np.float32)
        self._left = self._fingerprint()
        return out

    def reset(self):
        """forget the rows drawn ahead: the next draw comes from the generator as it is now"""
        if self._next is not None:
            self._next.result()
        self._buf, self._pos, self._next = np.zeros((0, self.Z), np.float32), 0, None
        self._left = self._fingerprint()

    def close(self):
        self._pool.shutdown(wait=True)

    def take(self, n):
        if (self._next is None or self._next.done()) and self._fingerprint() != self._left:     # the caller moved the generator: nothing buffered is valid
            self.reset()
        out, need = [], int(n)
        while need > 0:
            avail = len(self._buf) - self._pos
            if avail == 0:
                if self._next is None:
                    self._next = self._pool.submit(self._draw, max(self.BLOCK, need))
                self._buf, self._pos, self._next = self._next.result(), 0, None
                continue
            k = min(avail, need)
            out.append(self._buf[self._pos:self._pos + k])
            self._pos += k
            need -= k
        if self._next is None and len(self._buf) - self._pos < self.BLOCK // 2:
            self._next = self._pool.submit(self._draw, self.BLOCK)
        if not out:
            return np.zeros((0, self.Z), np.float32)
        return out[0] if len(out) == 1 else np.concatenate(out, 0)


class _Shared(object):
    """State shared by the three model views: spec, parameters, engines."""

    def __init__(self, spec: ModelSpec, dtype, seed, device):
        self.spec, self.dtype, self.seed, self.device = spec, dtype, seed, device
        self.layout = ParamLayout.build(spec)
        self.params_host = init_params(spec, seed)     # authoritative copy until an engine exists
        self.engine = None                # the training engine (sized for the caller's batch_size); owns the parameters
        self.infer = None                 # forward-only engine for chip-filling inference batches; shares parameters and streams
        self.infer_cap = int(os.environ.get("MVAE_INFER_BATCH", "2048"))
        self.pver = [0]                   # parameter version: bumped by every update / load (Engine._pver), survives engine regrowth
        self.rng = np.random.default_rng(seed + 1)
        self.teacher_force = False        # extra ground-truth inputs in the lists (no effect on the graph)
        self.next_teacher_force = False
        self.dp = None                    # dp.DataParallel: minibatches are sharded over its ranks inside fit / evaluate / predict

    def get_engine(self, batch=16, training=True):
        """The training engine, sized for ``batch`` windows on first use (callers pass their ``batch_size``, not the length of
        the song at hand, so that songs of different lengths do not rebuild it).  A larger request rebuilds it; parameters AND
        optimizer state (Adam moments, step count) move to the new engine - Keras keeps both across fit calls."""
        from .engine import Engine                      # imported lazily: needs the HIP library and a GPU
        need = max(int(batch), 16)
        if self.engine is None or self.engine.maxB < need:
            old = self.engine
            params = old.get_params() if old is not None else self.params_host
            opt = old.get_optimizer_state() if old is not None else None
            self.engine = self.infer = None             # (the forward-only engine views the old engine's parameter buffer)
            del old
            self.engine = Engine(self.spec, max_batch=need, dtype=self.dtype, device=self.device, seed=self.seed,
                                 training=True)
            self.engine.set_params(params)
            self.engine._pver = self.pver               # (the same parameters in a bigger engine: the version does not move)
            if opt is not None:
                self.engine.set_optimizer_state(opt)
        return self.engine

    def get_infer(self, n):
        """The forward-only engine for ``n`` windows at once (capped: MVAE_INFER_BATCH, default 2048, and a third of the free
        HBM).  encoder.predict / autoencoder.evaluate / predict / decoder.predict* run through it at a chip-filling batch whatever
        ``batch_size`` the caller passes: a recurrence occupies one CU per 16 windows, so the caller's 256 windows are 16 of 256
        CUs for a pass that has no optimizer step to wait for (results are per window: identical either way)."""
        import torch
        from .engine import Engine
        base = self.get_engine()
        want = min(Engine.pad16(max(int(n), 16)), max(self.infer_cap, 16))
        if self.infer is None or self.infer.maxB < want:
            have = self.infer.maxB if self.infer is not None else 0
            self.infer = None
            size = 256
            while size < want:
                size *= 2
            size = min(size, max(Engine.pad16(self.infer_cap), 16))
            free, _ = torch.cuda.mem_get_info(base.device)
            per = Engine.forward_bytes_per_window(self.spec, base.kind)
            size = max(16, min(size, int(free // 3 // per) // 16 * 16))
            size = max(size, min(have, want))
            self.infer = Engine(self.spec, max_batch=size, dtype=self.dtype, device=self.device, seed=self.seed, training=False,
                                share=base)
        return self.infer

    def engine_device(self):
        import torch
        return torch.device(self.device)

    def param_version(self):
        return self.pver[0]

    def current_params(self):
        return self.engine.get_params() if self.engine is not None else self.params_host

    def set_params(self, named):
        self.params_host = OrderedDict((k, np.asarray(v, np.float32)) for k, v in named.items())
        if self.engine is not None:
            self.engine.set_params(self.params_host)
        else:
            self.pver[0] += 1

    def epsilon(self, n):
        st = getattr(self, "_eps_stream", None)
        if st is None or st.rng is not self.rng or st.Z != self.spec.Z or st.std != float(self.spec.epsilon_std):
            if st is not None:
                st.close()
            st = self._eps_stream = EpsilonStream(self.rng, self.spec.Z, self.spec.epsilon_std)     # (a caller replaced the generator)
        return st.take(n)

    def epsilon_batches(self, n, batch_size):
        """(n, Z) draws in the order a loop over minibatches of ``batch_size`` windows takes them (Keras' predict / evaluate
        sample once per batch): internal batching changes nothing about the random stream"""
        if n <= 0:
            return np.zeros((0, self.spec.Z), np.float32)
        return np.concatenate([self.epsilon(min(n, lo + batch_size) - lo) for lo in range(0, n, batch_size)], 0)

    def world(self):
        dp = self.dp
        return dp if (dp is not None and dp.world > 1) else None

    def encode_windows(self, X, I, Vel, Held, eps, lo, hi, out, X_tm=None):
        """sampled z of windows [lo, hi) -> rows [lo, hi) of the device tensor ``out``, through the forward-only engine at its
        batch; under data parallelism every rank encodes a contiguous share and the rows are exchanged (one all-reduce of a
        buffer that is zero outside the rank's share: n x Z floats)."""
        from .dp import shard_bounds
        if hi <= lo:
            return
        dp = self.world()
        a0, b0 = (lo, hi) if dp is None else tuple(lo + v for v in shard_bounds(hi - lo, dp.world, dp.rank))
        eng = self.get_infer(b0 - a0)
        st = eng.stager()
        if dp is not None:
            out[lo:hi].zero_()
        for p in range(a0, b0, eng.maxB):
            q = min(b0, p + eng.maxB)
            B = st.stage(p, q, X=X, I=I, Vel=Vel, Held=Held, eps=eps[p:q], X_tm=X_tm)
            out[p:q].copy_(eng.encode(B))
        if dp is not None:
            dp.allreduce_sum(out[lo:hi])

    def z_heads(self):
        """(Keras output name, metric prefix) of the outputs behind the decoder's, in the reference's order (vae_definition.py:
        392-432): style classifier on z, signature head, classifiers on the notes / instrument outputs"""
        sp = self.spec
        return ([("composer_decoder", "style")] if sp.style else []) + ([("signature_decoder", "sig")] if sp.signature else []) + \
               ([("composer_decoder_at_notes", "cnotes")] if sp.comp_notes else []) + \
               ([("composer_decoder_at_instruments", "cinstr")] if sp.comp_instr else [])

    def head_names(self):
        """decoder outputs in the reference's order (vae_definition.py:341-351): notes, instrument, velocity, held, next"""
        sp = self.spec
        return (["notes"] + (["instr"] if sp.meta_instrument else []) + (["vel"] if sp.meta_velocity else []) +
                (["held"] if sp.meta_held else []) + (["next"] if sp.meta_next else []))


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
        """[X, I_tiled (n,V,ID), Vel (n,T,1)] or bare X without meta heads (reference vae_definition.py:797-808)"""
        sp = self._s.spec
        x = _as_list(x)
        i = 0
        X = x[i]
        i += 1
        I = x[i] if sp.meta_instrument else None
        i += int(sp.meta_instrument)
        V = x[i] if sp.meta_velocity else None
        i += int(sp.meta_velocity)
        D = x[i] if sp.meta_held else None
        return X, I, V, D

    def predict(self, x, batch_size=32, verbose=0, device=False):
        """Sampled z for every window (fresh epsilon per call, drawn per ``batch_size`` windows as Keras' predict loop does:
        reference vae_definition.py:498-502).  The encoder itself runs at a chip-filling internal batch (_Shared.get_infer), under
        data parallelism on this rank's share of the windows; z is read back once.  ``device=True``: no read-back and - until
        something needs the values - no encoder pass either: a deferred DeviceLatent for the history slot of the next fit /
        evaluate (the history pre-pass of reference vae_training.py:788-798, fused into that call)."""
        import torch
        X, I, Vel, Held = self._unpack(x)
        n = np.asarray(X).shape[0]
        eps = self._s.epsilon_batches(n, batch_size)
        if device:
            return DeviceLatent(shared=self._s, arrays=(X, I, Vel, Held), eps=eps)
        eng = self._s.get_engine()
        out = torch.zeros((n, self._s.spec.Z), dtype=torch.float32, device=eng.device)
        self._s.encode_windows(X, I, Vel, Held, eps, 0, n, out)
        res = out.cpu().numpy()
        if self._s.infer is not None:
            self._s.infer.check_pipeline()
        return res


class Decoder(_ModelView):
    prefixes = ("dec.",)
    name = "decoder"

    def _unpack(self, x):
        """[start, z, (ground truth), (history), instr_start, vel_start, held_start, next_start, (next ground truth)]
        (reference vae_definition.py:253-327, built by prepare_decoder_input :820-865)"""
        sp = self._s.spec
        x = _as_list(x)
        start, z = x[0], x[1]
        i = 2 + int(bool(self._s.teacher_force))           # (the ground-truth input feeds the unused readout state: SURVEY F9)
        hist = x[i] if sp.history else None
        i += int(sp.history)
        res = dict(start_notes=start, z=z, hist=hist)
        if sp.add_dim:
            res["add"] = x[i]
            i += 1
        for flag, key in ((sp.meta_instrument, "start_instr"), (sp.meta_velocity, "start_vel"), (sp.meta_held, "start_held"),
                          (sp.meta_next, "start_next")):
            if flag:
                res[key] = x[i]
                i += 1
        return res

    def _run(self, x, batch_size, want_probs=True):
        """decoder forward over all windows at the forward-only engine's batch (``batch_size`` is the caller's: results are per
        window); every rank decodes everything it is given - decoding is pure replicas"""
        sp = self._s.spec
        a = self._unpack(x)
        z = np.asarray(a.pop("z"))
        n = z.shape[0]
        eng = self._s.get_infer(n)
        outs, idxs = [], []
        for lo in range(0, n, eng.maxB):
            hi = min(n, lo + eng.maxB)
            B = hi - lo
            eng.stage_decoder_inputs(B, z=z[lo:hi], **{k: (None if v is None else np.asarray(v)[lo:hi]) for k, v in a.items()})
            eng.decode(B, want_probs=want_probs)
            if want_probs:
                outs.append(eng.outputs(B))
            idxs.append(eng.note_indices(B))
        eng.check_pipeline()
        return outs, np.concatenate(idxs, 0) if idxs else np.zeros((0, sp.T), np.uint8)

    def predict(self, x, batch_size=32, verbose=0):
        outs, _ = self._run(x, batch_size)
        res = [np.concatenate([o[h] for o in outs], 0) for h in self._s.head_names()]
        return res if len(res) > 1 else res[0]

    def predict_note_indices(self, x, batch_size=32):
        """Fused argmax decode: (n, T) uint8 note index per row, computed on the device without materialising the
        (n, T, D) probability tensor on the host (replaces decoder.predict + sample_vector 'argmax')."""
        return self._run(x, batch_size, want_probs=False)[1]


class Autoencoder(_ModelView):
    name = "autoencoder"
    prefetch = True          # fit: the next minibatch's host conversion runs on a worker thread beside the step being enqueued

    def __init__(self, shared, encoder: Encoder):
        super().__init__(shared)
        self._enc = encoder

    # ---- names -------------------------------------------------------------------------------------------
    def _decoder_outputs(self):
        return self._s.head_names()

    @property
    def metrics_names(self):
        """Keras 2.0.8 naming: repeated 'decoder_loss' / 'decoder_acc' per decoder output; the reference
        de-duplicates them itself (vae_training.py:172-187)."""
        k = len(self._decoder_outputs())
        zh = self._s.z_heads()
        if k == 1 and not zh:
            return ["loss", "acc"]
        names = ["loss"] + ["decoder_loss"] * k + [n + "_loss" for n, _ in zh]
        names += ["decoder_acc"] * k + [n + "_acc" for n, _ in zh]
        return names

    def _history_keys(self):
        outs = self._decoder_outputs()
        zh = self._s.z_heads()
        if len(outs) == 1 and not zh:
            return [("loss", "loss"), ("acc", "notes_acc")]
        keys = [("loss", "loss")]
        if len(outs) == 1:
            keys += [("decoder_loss", "notes_loss"), ("decoder_acc", "notes_acc")]
        else:
            for i, o in enumerate(outs, 1):
                keys += [("decoder_loss_%d" % i, o + "_loss"), ("decoder_acc_%d" % i, o + "_acc")]
        for n, m in zh:
            keys += [(n + "_loss", m + "_loss"), (n + "_acc", m + "_acc")]
        return keys

    # ---- list unpacking (orders of reference vae_definition.py:924-1040) -------------------------------------
    def _unpack_x(self, x):
        """[X, Y_start, (Y ground truth), H, instr_start, I, vel_start, V, held_start, D, next_start, (next ground truth)]
        (reference vae_definition.py:924-1028) -> dict of Stager.stage keyword arguments"""
        sp = self._s.spec
        x = _as_list(x)
        a = dict(X=x[0], start_notes=x[1])
        i = 2 + int(bool(self._s.teacher_force))
        a["hist"] = x[i] if sp.history else None
        i += int(sp.history)
        if sp.add_dim:
            a["Add"] = x[i]
            i += 1
        for flag, k_start, k_in in ((sp.meta_instrument, "start_instr", "I"), (sp.meta_velocity, "start_vel", "Vel"),
                                    (sp.meta_held, "start_held", "Held")):
            if flag:
                a[k_start], a[k_in] = x[i], x[i + 1]
                i += 2
        if sp.meta_next:
            a["start_next"] = x[i]
        return a

    def _unpack_y(self, y):
        """[Y, I, V, D, N, C] (reference vae_definition.py:926-1031) -> targets as Stager.stage keyword arguments (the instrument,
        velocity and held-notes targets are the encoder inputs themselves)"""
        sp = self._s.spec
        y = _as_list(y)
        i = 1 + int(sp.meta_instrument) + int(sp.meta_velocity) + int(sp.meta_held)
        t = dict(Y=y[0])
        if sp.meta_next:
            t["Next"] = y[i]
            i += 1
        for _, m in self._s.z_heads():                       # [C, S, C, C]
            t["S" if m == "sig" else "C_"] = y[i]
            i += 1
        return t

    def _unpack_w(self, w, n):
        """Keras maps a ``sample_weight`` LIST to the model's outputs BY POSITION: [notes, instrument, velocity, held, next,
        style].  (The reference builds its list as notes / style / ... / instrument / velocity / held / next,
        vae_definition.py:930-1028 - every entry but the first is all-ones of shape (n,), so the mismatch is invisible there; it
        is applied here as Keras would apply it.)  Returns Stager.stage keyword arguments; all-ones weights are dropped."""
        sp = self._s.spec
        if w is None:
            return {}
        w = list(w) if isinstance(w, (list, tuple)) else [w]
        outs = self._s.head_names() + [m for _, m in self._s.z_heads()]
        if len(w) != len(outs):
            raise ValueError("sample_weight has %d entries for the %d outputs %s" % (len(w), len(outs), outs))
        res = {}
        for k, v in zip(outs, w):
            v = np.asarray(v)
            want = (n, sp.T) if k == "notes" else (n,)
            if v.shape != want:
                raise ValueError("sample_weight of output %r must have shape %s (the notes output uses sample_weight_mode="
                                 "'temporal'), got %s" % (k, want, v.shape))
            if not np.all(v == 1):
                res["w_" + k] = v
        return res

    def _dp(self, dp):
        dp = dp if dp is not None else self._s.dp
        return dp if (dp is not None and dp.world > 1) else None

    @staticmethod
    def _history_source(hist):
        if isinstance(hist, DeviceLatent):
            return None, hist.z             # (runs the pre-pass if it is still deferred)
        return hist, None

    # ---- Keras methods -----------------------------------------------------------------------------------------
    def fit(self, x, y, epochs=1, batch_size=32, shuffle=False, sample_weight=None, verbose=0, dp=None, allreduce=None):
        """One optimizer step per minibatch of <= batch_size consecutive windows (reference vae_training.py:804-809).  History
        values are batch-size-weighted means over the minibatches of each epoch (Keras BaseLogger semantics), accumulated on
        the device and read back ONCE per epoch.

        A deferred DeviceLatent in the history slot (``encoder.predict(..., device=True)`` on the same song, reference
        vae_training.py:788-798) is the FUSED history pre-pass: the windows of the later minibatches are encoded first - forward
        only, at a chip-filling batch, with the weights this call starts from - and the first minibatch's history comes out of its
        own encoder forward (Engine.train_step_begin ``hist_fused``): a song of one minibatch runs its encoder once, not twice.

        Data parallel (``dp``: dp.DataParallel, or VAE.set_data_parallel): every rank calls fit with the SAME arguments; each
        global minibatch is split contiguously over the ranks (dp.shard_bounds), the losses are normalised by the GLOBAL counts,
        gradients are summed by one all-reduce per step - the parameters after every step equal the single-process run's (to
        f32 summation order), ragged last minibatches and empty shards included, and every rank executes the same number of
        collectives by construction.  NOTE: the ranks SPLIT ``batch_size``; keep per-GPU work constant by growing it with the
        world (256 windows per rank fill 16 of a GPU's 256 CUs per recurrence - fewer do not run faster).  ``allreduce``
        (legacy hook: each rank trains on its OWN minibatches, mean of gradients) is kept for bench-style callers."""
        if shuffle:
            raise NotImplementedError("shuffle=True (the reference always passes shuffle=False)")
        from .dp import shard_bounds
        from .staging import Norm
        sp = self._s.spec
        a = self._unpack_x(x)
        a.update(self._unpack_y(y))
        n = np.asarray(a["X"]).shape[0]
        ws = self._unpack_w(sample_weight, n)
        a.update(ws)
        dp = self._dp(dp)
        eng = self._s.get_engine(batch_size, training=True)
        eng.status_allreduce = dp.allreduce_max if dp is not None else None
        st = eng.stager()
        lat = a.get("hist") if isinstance(a.get("hist"), DeviceLatent) else None
        fused = None
        if lat is not None and lat.deferred:
            lat.check_version()
            if dp is None and epochs == 1 and n > 0 and lat.matches(a["X"], n, a.get("I"), a.get("Vel"), a.get("Held")):
                fused = lat
                b0 = min(n, batch_size)
                if n > b0:      # histories of the later minibatches: before the first update, forward only, chip-filling; their
                    from .staging import host_onehot_to_index_tm      # windows are converted once for this pass and their train steps
                    xtm = host_onehot_to_index_tm(a["X"], b0, n) if not sp.attach else None
                    a["X_tm"] = (xtm, b0) if xtm is not None else None
                    self._s.encode_windows(a["X"], a.get("I"), a.get("Vel"), a.get("Held"), lat._root().eps, b0, n, lat._root()._z,
                                           X_tm=a["X_tm"])
                a["hist"], a["hist_dev"] = None, lat._root()._z
        if fused is None:
            a["hist"], a["hist_dev"] = self._history_source(a.get("hist"))
        keys = self._history_keys()
        pending = []
        # Minibatch i+1 is CONVERTED (the host packers: 2-6 ms per 256 windows of float64 one-hot rows) on a worker thread while the
        # host enqueues step i - a call in which the paced host mostly waits for the device (Stager.prefetch).  One draw of epsilon
        # per GLOBAL minibatch, in minibatch order as before: the next minibatch's draw just happens one step early.
        order = [(e, lo) for e in range(epochs) for lo in range(0, n, batch_size)]

        def plan(i):
            e, lo = order[i]
            hi = min(n, lo + batch_size)
            eps = self._s.epsilon(hi - lo)               # one draw per GLOBAL minibatch: every rank holds the same stream
            a0, b0 = (0, hi - lo) if dp is None else shard_bounds(hi - lo, dp.world, dp.rank)
            first = fused is not None and lo == 0
            kw = None
            if b0 > a0:
                kw = dict(eps=eps[a0:b0], norm=Norm.of(lo, hi, sp.T, **ws), defer_targets=True,
                          eps2=fused._root().eps[:hi] if first else None, **a)
            return dict(e=e, lo=lo, hi=hi, a0=a0, b0=b0, first=first, kw=kw, prefetched=False)

        cur = plan(0) if order else None
        acc = None
        for i in range(len(order)):
            b, cur = cur, None
            lo, hi, a0, b0, first = b["lo"], b["hi"], b["a0"], b["b0"], b["first"]
            if lo == 0:
                acc = eng.reset_accumulated()
            if b["kw"] is not None:
                # the encoder's inputs first; the heads' targets are converted while the encoder already runs on the device
                B = st.stage(lo + a0, lo + b0, prefetched=b["prefetched"], **b["kw"])
                eng.train_step_begin(B, hist_fused=(eng._v("in.eps2", eng.pad16(B), sp.Z), fused._root()._z[:hi]) if first else None)
                st.finish_targets()
                if i + 1 < len(order):
                    cur = plan(i + 1)
                    if cur["kw"] is not None and not cur["first"] and self.prefetch:
                        st.prefetch(cur["lo"] + cur["a0"], cur["lo"] + cur["b0"], **cur["kw"])
                        cur["prefetched"] = True
                eng.train_step_finish(B, allreduce=dp.hook(eng) if dp is not None else allreduce)
                eng.accumulate_metrics(hi - lo)
                if first:
                    fused.mark_valid()
            else:
                eng.train_step_empty(dp.hook(eng))
            if cur is None and i + 1 < len(order):
                cur = plan(i + 1)
            if hi < n:
                continue
            if dp is not None:          # every rank, every fit call, here - not when (and if) a rank reads the History
                dp.allreduce_sum(acc)
            pending.append(acc)

        def resolve(out):
            for acc_e in pending:
                m = eng.read_accumulated(n, acc=acc_e)
                for k, src in keys:
                    out.setdefault(k, []).append(m[src])

        history = History(resolve)
        history.epoch = list(range(epochs))
        return history

    def _forward_all(self, x, y, batch_size, want_probs):
        """forward (+ losses) over all windows at the forward-only engine's batch; epsilon drawn per ``batch_size`` windows as
        Keras' evaluate / predict loops draw it.  With targets and data parallelism every rank takes a contiguous share of each
        internal batch (global normalisers, accumulators summed over the ranks: reference vae_training.py:286-300 on all GPUs)."""
        from .dp import shard_bounds
        from .staging import Norm
        sp = self._s.spec
        a = self._unpack_x(x)
        if y is not None:
            a.update(self._unpack_y(y))
        n = np.asarray(a["X"]).shape[0]
        a["hist"], a["hist_dev"] = self._history_source(a.get("hist"))
        dp = self._s.world() if (y is not None and not want_probs) else None
        eps = self._s.epsilon_batches(n, batch_size)
        eng = self._s.get_infer(n if dp is None else -(-n // dp.world))
        st = eng.stager()
        outs = []
        eng.reset_accumulated()
        step = eng.maxB * (dp.world if dp is not None else 1)
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            a0, b0 = (0, hi - lo) if dp is None else shard_bounds(hi - lo, dp.world, dp.rank)
            if b0 <= a0:
                continue
            B = st.stage(lo + a0, lo + b0, eps=eps[lo + a0:lo + b0], norm=Norm.of(lo, hi, sp.T) if y is not None else None, **a)
            if y is None:
                eng._have_targets = False
                eng.scal.zero_()
                if eng._weights_dirty:
                    eng.prepare_weights()
                eng.encoder_forward(B, with_init=True)
                eng.decoder_forward(B, want_probs=True)
            else:
                eng.eval_step(B, want_probs=want_probs)
                eng.accumulate_metrics(hi - lo)
            if want_probs:
                outs.append(eng.outputs(B))
        tot = None
        if y is not None:
            tot = eng.read_accumulated(n, allreduce_sum=dp.allreduce_sum if dp is not None else None)
        else:
            eng.check_pipeline()
        return tot, outs, n

    def evaluate(self, x, y, batch_size=32, verbose=0, sample_weight=None):
        """Forward + losses over the song's minibatches (reference vae_training.py:300); the list is aligned with
        ``metrics_names``."""
        if sample_weight is not None:
            raise NotImplementedError("evaluate(sample_weight=...): the reference never passes it (vae_training.py:300)")
        m, _, n = self._forward_all(x, y, batch_size, want_probs=False)
        pairs = self._history_keys()
        losses = [src for k, src in pairs if "loss" in k]
        accs = [src for k, src in pairs if "acc" in k]
        return [m[k] for k in losses + accs]

    def predict(self, x, batch_size=32, verbose=0):
        sp = self._s.spec
        _, outs, _ = self._forward_all(x, None, batch_size, want_probs=True)
        res = [np.concatenate([o[h] for o in outs], 0) for h in self._s.head_names() + [m for _, m in self._s.z_heads()]]
        return res if len(res) > 1 else res[0]


class VAE(object):
    """reference vae_definition.py:39.  ``create`` takes the reference's keyword arguments unchanged; two extra,
    optional ones select the arithmetic (``compute_dtype`` 'bf16' | 'f32') and the initialiser seed."""

    def __init__(self):
        self.encoder = self.decoder = self.autoencoder = self.composer_decoder = None

    def create(self, compute_dtype="bf16", seed=0, device="cuda:0", attach_dim=0, **kw):
        """``attach_dim``: settings.instrument_dim when ``attach_instruments`` is on (the reference's create call does not carry
        it: its rows are then pitch one-hot | instrument one-hot, input_dim = output_dim = notes + silent + attach_dim)"""
        for k, v in kw.items():
            setattr(self, k, v)
        self.spec = spec_from_create_kwargs(dict(kw, attach_dim=attach_dim))
        shared = _Shared(self.spec, compute_dtype, seed, device)
        shared.teacher_force = bool(kw.get("teacher_force", False))
        shared.next_teacher_force = bool(kw.get("meta_next_notes_teacher_force", False))
        self._shared = shared
        self.encoder = Encoder(shared)
        self.decoder = Decoder(shared)
        self.autoencoder = Autoencoder(shared, self.encoder)
        self.composer_decoder = None     # parameter-free softmax over z[:, :C]; reported through the autoencoder
        return self

    def set_data_parallel(self, dp):
        """dp.DataParallel (or None): ``autoencoder.fit`` shards every minibatch over its ranks from now on."""
        self._shared.dp = dp
        return self
