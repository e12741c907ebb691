"""Input staging of the device engine: the caller's host arrays -> ONE pinned block -> ONE asynchronous upload.

What the reference's callers hand to ``autoencoder.fit`` / ``evaluate`` / ``predict`` (reference vae_training.py:802-809,
lists built by vae_definition.py:880-1045) are float64 NumPy windows: one-hot note rows (n, T, 61), one-hot instrument
rows (n, V, 16), velocity (n, T, 1), the history latent (n, Z), one-hot style targets, sample weights.  The engine consumes
one byte per row, time-major, padded to 16 windows, plus f32 side inputs.  ``Stager.stage`` converts one minibatch
[lo, hi) of such a song with the multi-threaded host packers of the C ABI (csrc/hostpack.cpp: validation + argmax +
transpose + padding in one pass over the caller's array), writes everything into a pinned mirror of the engine's contiguous
input block and uploads it with a single stream-ordered copy.  Two pinned mirrors alternate, so the host prepares minibatch
k+1 while the device still runs step k; nothing here synchronises the device.

Data parallelism: every Keras loss of this graph is a (weighted) batch mean.  ``norm`` carries the
GLOBAL minibatch size and non-zero weight counts, so that a rank staging only its shard of the minibatch produces its
share of the global-mean gradient: summing over ranks (all-reduce) gives exactly the single-process gradient, ragged
shards included ("sum, then divide by the global count").
"""
from __future__ import annotations

import ctypes as C
import queue
import threading

import numpy as np
import torch

from . import hiplib as hl

_KIND = {np.dtype(np.float64): hl.HOST_F64, np.dtype(np.float32): hl.HOST_F32, np.dtype(np.uint8): hl.HOST_U8}


class Norm(object):
    """Normalisers of one GLOBAL minibatch (Keras weighted objectives, SURVEY Appendix A.7: score*w / mean(w != 0), then
    the mean over the remaining axes): B windows, and the number of non-zero sample weights per output."""

    def __init__(self, B, nz_notes, nz_instr, nz_vel, nz_style, nz_held=None, nz_next=None, nz_sig=None, nz_cnotes=None,
                 nz_cinstr=None):
        self.B, self.nz_notes, self.nz_instr, self.nz_vel, self.nz_style = int(B), nz_notes, nz_instr, nz_vel, nz_style
        self.nz_held = int(B) if nz_held is None else nz_held
        self.nz_next = int(B) if nz_next is None else nz_next
        self.nz_sig = int(B) if nz_sig is None else nz_sig
        self.nz_cnotes = int(B) if nz_cnotes is None else nz_cnotes
        self.nz_cinstr = int(B) if nz_cinstr is None else nz_cinstr

    @staticmethod
    def of(lo, hi, T, w_notes=None, w_instr=None, w_vel=None, w_style=None, w_held=None, w_next=None, w_sig=None, w_cnotes=None,
           w_cinstr=None):
        B = hi - lo

        def nz(w, full):
            if w is None:
                return full
            c = int(np.count_nonzero(np.asarray(w)[lo:hi]))
            return max(c, 1)

        return Norm(B, nz(w_notes, B * T), nz(w_instr, B), nz(w_vel, B), nz(w_style, B), nz(w_held, B), nz(w_next, B),
                    nz(w_sig, B), nz(w_cnotes, B), nz(w_cinstr, B))


def _host_kind(a):
    k = _KIND.get(a.dtype)
    if k is None:
        raise TypeError("host arrays must be float64, float32 or uint8 (got %s)" % a.dtype)
    return k


def _c(a):
    a = np.asarray(a)
    if a.dtype not in _KIND:
        a = a.astype(np.float64)
    return a if a.flags.c_contiguous else np.ascontiguousarray(a)


def host_onehot_to_index(a, what="input"):
    """(n, T, K) one-hot rows (float64 / float32 / uint8) -> (n, T) uint8 indices through the C ABI's host packer; raises
    NotImplementedError for anything that is not exactly one 1 among zeros per row (no device needed)."""
    a = _c(a)
    if a.ndim != 3:
        raise ValueError("%s: expected (n, T, K) one-hot rows, got %s" % (what, a.shape))
    n, T, K = a.shape
    out = np.empty((T, max(n, 1)), np.uint8)
    bad = C.c_int64(-1)
    rc = hl.load().mvae_host_onehot_to_index_tm(a.ctypes.data, _host_kind(a), n, T, K, 0, n, out.ctypes.data, max(n, 1), 0,
                                                C.byref(bad))
    if rc == hl.E_FORMAT:
        raise NotImplementedError("%s must be one-hot rows of width <= 255 (reference layout, import_midi.py:255-262): row %d of "
                                  "window %d is not; dense / multi-hot rows need the dense input projection, which is not built.  (A DECODED song - "
                                  "process_decoder_outputs: no silent column, all-zero rows where the model chose silence - becomes windows "
                                  "again through import_midi.windows_from_unrolled_rolls, as in the reference.)"
                                  % (what, bad.value % T, bad.value // T))
    hl.check(rc, "mvae_host_onehot_to_index_tm")
    return np.ascontiguousarray(out[:, :n].T)


def host_onehot_to_index_tm(a, lo, hi, what="notes input"):
    """windows [lo, hi) of (n, T, K) one-hot rows -> (T, hi - lo) uint8 indices, TIME-MAJOR (the layout the engine stages): the
    conversion of a song's windows done ONCE when two passes read them - the history pre-pass and the train steps of the later
    minibatches (Stager.stage ``X_tm``).  None if ``a`` is not an array of one-hot rows (already indices)."""
    a = _c(a)
    if a.ndim != 3 or hi <= lo:
        return None
    n, T, K = a.shape
    out = np.empty((T, hi - lo), np.uint8)
    bad = C.c_int64(-1)
    rc = hl.load().mvae_host_onehot_to_index_tm(a.ctypes.data, _host_kind(a), n, T, K, lo, hi, out.ctypes.data, hi - lo, 0,
                                                C.byref(bad))
    if rc == hl.E_FORMAT:
        raise NotImplementedError("%s must be one-hot rows (reference layout, import_midi.py:255-262): row %d of window %d is not"
                                  % (what, bad.value % T, bad.value // T))
    hl.check(rc, "mvae_host_onehot_to_index_tm")
    return out


class Stager(object):
    def __init__(self, engine):
        self.eng = engine
        self.lib = hl.load()
        self.regions = engine._in_regions            # name -> (byte offset, byte length, torch dtype)
        nbytes = engine._in_block.numel()
        self.host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.np_host = [h.numpy() for h in self.host]
        for h in self.np_host:
            h[:] = 0
        self.done = [None, None]                     # event after the last upload from each mirror
        self.k = 0
        self._late = None
        self._pf = None                              # a minibatch being converted ahead on the worker thread (prefetch)
        self._worker, self._jobs = None, None
        self._notes = []                             # (name, rows) of the start rows a conversion wrote: Engine._note_start at commit

    def _view(self, k, name, dtype, n):
        off, length, _ = self.regions[name]
        item = np.dtype(dtype).itemsize
        assert n * item <= length, (name, n, length)
        return self.np_host[k][off:off + n * item].view(dtype)

    # ---- converters (write into the pinned mirror) ----------------------------------------------------------------
    def _rows_u8(self, k, name, arr, lo, hi, steps, width, Bp, fill, what):
        """one-hot (n, steps, width) or index (n, steps) rows -> (steps, Bp) uint8 time-major"""
        a = _c(arr)
        out = self._view(k, name, np.uint8, steps * Bp)
        if a.ndim == 3 or (a.ndim == 2 and steps == 1 and a.shape[1] == width and a.dtype != np.uint8):
            if a.shape[-1] != width or a.size != a.shape[0] * steps * width:
                raise ValueError("%s: expected (n, %d, %d) one-hot rows, got %s" % (what, steps, width, a.shape))
            bad = C.c_int64(-1)
            rc = self.lib.mvae_host_onehot_to_index_tm(a.ctypes.data, _host_kind(a), a.shape[0], steps, width, lo, hi,
                                                       out.ctypes.data, Bp, fill, C.byref(bad))
            if rc == hl.E_FORMAT:
                raise NotImplementedError(
                    "%s must be one-hot rows (reference layout, import_midi.py:255-262): row %d of window %d is not (rows with "
                    "an attached instrument one-hot - attach_instruments - need VAE.create(attach_dim=settings.instrument_dim); other "
                    "dense / multi-hot rows are not built)" % (what, bad.value % steps, bad.value // steps))
            hl.check(rc, "mvae_host_onehot_to_index_tm")
        else:
            a = a.reshape(a.shape[0], -1)
            if a.dtype != np.uint8 or a.shape[1] != steps:
                raise ValueError("%s: expected (n, %d) uint8 indices or one-hot rows, got %s %s" % (what, steps, a.shape, a.dtype))
            hl.check(self.lib.mvae_host_index_to_tm(a.ctypes.data, a.shape[0], steps, lo, hi, out.ctypes.data, Bp, fill),
                     "mvae_host_index_to_tm")

    def _rows_twohot(self, k, name1, name2, arr, lo, hi, steps, width, attach, Bp, fill, what, absolute):
        """two-hot (n, steps, width) rows - pitch one-hot | attached instrument one-hot (attach_instruments) -> two (steps, Bp)
        uint8 index rolls; ``absolute``: the second index as a column of the full row (targets) instead of within its block"""
        a = _c(arr)
        if a.ndim != 3 or a.shape[-1] != width or a.shape[1] != steps:
            raise ValueError("%s: expected (n, %d, %d) two-hot rows, got %s" % (what, steps, width, a.shape))
        o1, o2 = self._view(k, name1, np.uint8, steps * Bp), self._view(k, name2, np.uint8, steps * Bp)
        bad = C.c_int64(-1)
        rc = self.lib.mvae_host_twohot_to_index_tm(a.ctypes.data, _host_kind(a), a.shape[0], steps, width, width - attach, lo, hi,
                                                   o1.ctypes.data, o2.ctypes.data, Bp, fill, C.byref(bad))
        if rc == hl.E_FORMAT:
            raise NotImplementedError("%s must be pitch one-hot | instrument one-hot rows (attach_instruments with a 1hot "
                                      "instrument_attach_method, reference import_midi.py:288-292): row %d of window %d is not"
                                      % (what, bad.value % steps, bad.value // steps))
        hl.check(rc, "mvae_host_twohot_to_index_tm")
        if absolute:
            o2r = o2.reshape(steps, Bp)
            o2r[:, :hi - lo] += np.uint8(width - attach)

    def _rows_f32(self, k, name, arr, lo, hi, steps, Bp, scale=1.0):
        a = _c(arr)
        a = a.reshape(a.shape[0], -1)
        if a.shape[1] != steps:
            raise ValueError("%s: expected (n, %d) values, got %s" % (name, steps, a.shape))
        out = self._view(k, name, np.float32, steps * Bp)
        hl.check(self.lib.mvae_host_rows_to_tm_f32(a.ctypes.data, _host_kind(a), a.shape[0], steps, lo, hi, float(scale),
                                                   out.ctypes.data, Bp), "mvae_host_rows_to_tm_f32")

    def _per_window(self, k, name, w, lo, hi, steps, Bp, scale):
        """per-window weight (n,) (None = ones) -> (steps, Bp) f32 = scale * w repeated over the steps, pad columns 0"""
        B = hi - lo
        out = self._view(k, name, np.float32, steps * Bp).reshape(steps, Bp)
        if w is None:
            out[:, :B] = scale
        else:
            out[:, :B] = (np.asarray(w, np.float64)[lo:hi] * scale).astype(np.float32)[None, :]
        out[:, B:] = 0.0

    def _rows_bm(self, k, name, arr, lo, hi, width, Bp):
        """(n, width) -> first B rows of (Bp, width) f32, pad rows zero; None = zeros"""
        B = hi - lo
        out = self._view(k, name, np.float32, Bp * width).reshape(Bp, width)
        if arr is None:
            out[:] = 0.0
        else:
            out[:B] = np.asarray(arr).reshape(-1, width)[lo:hi]
            out[B:] = 0.0
        if name.startswith("in.start_"):
            self._notes.append((name, None if arr is None else out[:B]))

    # ---- one minibatch ------------------------------------------------------------------------------------------
    def stage(self, lo, hi, *, X, I=None, Vel=None, eps=None, hist=None, hist_dev=None, z=None, Y=None, C_=None,
              start_notes=None, start_instr=None, start_vel=None, w_notes=None, w_instr=None, w_vel=None, w_style=None,
              norm=None, batch_local=False, Held=None, Next=None, start_held=None, start_next=None, w_held=None, w_next=None,
              Add=None, S=None, w_sig=None, w_cnotes=None, w_cinstr=None, defer_targets=False, eps2=None, X_tm=None,
              prefetched=False, _prefetch=False):
        """Windows [lo, hi) of a song -> the engine's input block (asynchronous).  Arrays are whole-song arrays indexed by
        window unless ``batch_local`` (then they hold exactly the hi-lo windows of this batch and lo is an offset of 0).
        ``eps`` is always batch-local (B, Z), already scaled by epsilon_std.  ``hist``: host (n, Z) history rows; ``hist_dev``:
        a DEVICE tensor (n, Z) of sampled z whose row i-1 is the history of window i (zeros for window 0) - the fused history
        pre-pass; neither: zeros.  ``eps2`` (B, Z): the draw of a history pre-pass FUSED into this train step (the engine then
        writes the history columns itself: Engine.train_step_begin).  ``X_tm`` = (indices (T, m) uint8, first window): the notes
        input of windows [first, first + m) already converted (host_onehot_to_index_tm) - used instead of X where it covers
        [lo, hi).  Returns the number of windows staged.

        ``defer_targets``: convert and upload everything the ENCODER needs now and leave the decoder heads' targets and row weights
        (the second 64 MB float64 tensor of a minibatch) to ``finish_targets()`` - called after the encoder's launches are
        enqueued, so that conversion runs on the host while the encoder recurrences run on the device.

        ``prefetched``: this minibatch was handed to ``prefetch()`` (same arguments) while the previous step was being enqueued:
        its conversion - both halves - ran on a worker thread beside the paced host's waits; only the uploads are left."""
        eng, s = self.eng, self.eng.spec
        # what a prefetched conversion was made FROM: the very objects (and the scalars that decide the conversion).  A stage()
        # call takes it only for the same ones - another song of equal length, an epsilon drawn again after the prefetch or other
        # row weights convert afresh (window range and mirror alone used to decide)
        token = (lo, hi, bool(batch_local), bool(defer_targets)) + tuple(
            id(a) for a in (X, I, Vel, eps, hist, hist_dev, z, Y, C_, start_notes, start_instr, start_vel, w_notes, w_instr, w_vel,
                            w_style, norm, Held, Next, start_held, start_next, w_held, w_next, Add, S, w_sig, w_cnotes, w_cinstr,
                            eps2, X_tm))
        if batch_local:
            lo, hi = 0, hi - lo
        B = hi - lo
        if B <= 0 or B > eng.maxB:
            raise ValueError("batch of %d windows does not fit the engine (max %d)" % (B, eng.maxB))
        Bp = eng.pad16(B)
        T, V = s.T, s.V
        k = self.k
        have_targets = Y is not None
        nm = (norm if norm is not None else Norm.of(lo, hi, T, w_notes, w_instr, w_vel, w_style, w_held, w_next, w_sig, w_cnotes,
                                                    w_cinstr)) if have_targets else None

        def convert(late_now):
            """everything that is HOST work: the caller's arrays -> mirror k.  Returns (start rows written, the deferred half of the
            conversion - the decoder heads' targets - or None when it ran too)"""
            if self.done[k] is not None:
                self.done[k].synchronize()              # the upload that last read this mirror has completed
            self._notes = []
            if s.attach:
                self._rows_twohot(k, "in.x_idx", "in.xa_idx", X, lo, hi, T, s.Din, s.attach, Bp, 0, "notes input", False)
            elif X_tm is not None and not batch_local and X_tm[1] <= lo and hi <= X_tm[1] + X_tm[0].shape[1]:
                out = self._view(k, "in.x_idx", np.uint8, T * Bp).reshape(T, Bp)
                out[:, :B] = X_tm[0][:, lo - X_tm[1]:hi - X_tm[1]]
                out[:, B:] = 0
            else:
                self._rows_u8(k, "in.x_idx", X, lo, hi, T, s.Din, Bp, 0, "notes input")
            if eng.enc_bi:          # the backward RNNs of a bidirectional encoder read the roll reversed in time
                self._view(k, "in.x_idx_rev", np.uint8, T * Bp).reshape(T, Bp)[:] = self._view(k, "in.x_idx", np.uint8, T * Bp).reshape(T, Bp)[::-1]
            if s.meta_instrument:
                self._rows_u8(k, "in.i_idx", I, lo, hi, V, s.ID, Bp, 0, "instrument input")
            if s.meta_velocity:
                self._rows_f32(k, "in.vel", Vel, lo, hi, T, Bp)
            if s.meta_held:
                self._rows_u8(k, "in.d_idx", Held, lo, hi, T, 2, Bp, 0, "held-notes input")
                self._rows_bm(k, "in.start_held", start_held, lo, hi, 2, Bp)
            if s.meta_next:
                self._rows_bm(k, "in.start_next", start_next, lo, hi, s.Dout, Bp)
            self._rows_bm(k, "in.eps", eps, 0, B, s.Z, Bp)
            if eps2 is not None:
                self._rows_bm(k, "in.eps2", eps2, 0, B, s.Z, Bp)
            self._rows_bm(k, "in.start_notes", start_notes, lo, hi, s.Dout, Bp)
            if s.meta_instrument:
                self._rows_bm(k, "in.start_instr", start_instr, lo, hi, s.ID, Bp)
            if s.meta_velocity:
                self._rows_bm(k, "in.start_vel", start_vel, lo, hi, 1, Bp)
            if s.history and hist_dev is None and eps2 is None:
                self._rows_bm(k, "in.hist", hist, lo, hi, s.Z, Bp)
            if z is not None:
                self._rows_bm(k, "in.z", z, lo, hi, s.Z, Bp)
            if s.add_dim:
                self._rows_bm(k, "in.add", Add, lo, hi, s.add_dim, Bp)
            late = None
            if have_targets:
                def late():
                    if s.attach:
                        self._rows_twohot(k, "in.y_idx", "in.ya_idx", Y, lo, hi, T, s.Dout, s.attach, Bp, 255, "notes target", True)
                    else:
                        self._rows_u8(k, "in.y_idx", Y, lo, hi, T, s.Dout, Bp, 255, "notes target")
                    if w_notes is None:
                        out = self._view(k, "in.rw_notes", np.float32, T * Bp).reshape(T, Bp)
                        out[:, :B] = 1.0 / nm.nz_notes
                        out[:, B:] = 0.0
                    else:
                        self._rows_f32(k, "in.rw_notes", w_notes, lo, hi, T, Bp, scale=1.0 / nm.nz_notes)
                    if s.meta_instrument:
                        self._per_window(k, "in.rw_instr", w_instr, lo, hi, V, Bp, 1.0 / (nm.nz_instr * V))
                    if s.meta_velocity:
                        self._per_window(k, "in.rw_vel", w_vel, lo, hi, T, Bp, 1.0 / (nm.nz_vel * T))
                    if s.meta_held:
                        self._per_window(k, "in.rw_held", w_held, lo, hi, T, Bp, 1.0 / (nm.nz_held * T))
                    if s.meta_next:
                        self._per_window(k, "in.rw_next", w_next, lo, hi, T, Bp, 1.0 / (nm.nz_next * T))
                        self._rows_u8(k, "in.n_idx", Next, lo, hi, T, s.Dout, Bp, 255, "next-notes target")

                if late_now:
                    late()
                    late = None
                if s.style:
                    self._per_window(k, "in.rw_style", w_style, lo, hi, 1, Bp, 1.0 / nm.nz_style)
                if s.style or s.comp_notes or s.comp_instr:
                    self._rows_u8(k, "in.c_idx", C_, lo, hi, 1, s.C, Bp, 255, "style target")
                if s.signature:
                    self._rows_bm(k, "in.sig", S, lo, hi, s.SD, Bp)
                    self._per_window(k, "in.rw_sig", w_sig, lo, hi, 1, Bp, 1.0 / nm.nz_sig)
                if s.comp_notes:
                    self._per_window(k, "in.rw_cnotes", w_cnotes, lo, hi, 1, Bp, 1.0 / nm.nz_cnotes)
                if s.comp_instr:
                    self._per_window(k, "in.rw_cinstr", w_cinstr, lo, hi, 1, Bp, 1.0 / nm.nz_cinstr)
            return self._notes, late

        if _prefetch:               # (prefetch(): both halves on the worker thread, nothing of the engine touched)
            box = {"done": threading.Event()}

            def work():
                try:
                    box["out"] = convert(True)
                except BaseException as e:      # (raised by the stage() call that takes the minibatch)
                    box["err"] = e
                box["done"].set()
            self._pf = (box["done"], k, token, box)
            self._worker_submit(work)
            return B
        self.k ^= 1
        self._late = None
        pf, self._pf = self._pf, None
        if pf is not None:
            pf[0].wait()
            if "err" in pf[3] and prefetched and pf[2] == token:
                raise pf[3]["err"]
        if prefetched and pf is not None and pf[1] == k and pf[2] == token and "out" in pf[3]:
            notes, late = pf[3]["out"]
        else:
            notes, late = convert(not defer_targets)
        for name, val in notes:
            eng._note_start(name, val)
        self._late = (k, late) if (late is not None) else ((k, None) if (have_targets and defer_targets) else None)
        eng.norm_B = float(nm.B) if have_targets else float(B)
        # ---- ONE upload, ordered on the current stream behind whatever still reads the block -------------------------------
        cut = eng._in_late_off if (have_targets and defer_targets) else None
        if cut is None:
            self._late = None
            eng._in_block.copy_(self.host[k], non_blocking=True)
        else:
            eng._in_block[:cut].copy_(self.host[k][:cut], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[k] = ev
        # [z | history]: the decoder's initial-state Denses read one (Bp, zin) operand
        zh = eng._v("zh", Bp, s.zin)
        from . import ops
        if s.history and eps2 is None:
            if hist_dev is not None:
                # history of window i = sampled z of window i-1, zeros for the first window of the song
                first = 1 if lo == 0 else 0
                ops.copy2d(zh[:, s.Z:], hist_dev, B, s.Z, src_row0=lo - 1, zero_rows=first)
                if Bp > B:
                    zh[B:, s.Z:].zero_()
            else:
                ops.copy2d(zh[:, s.Z:], eng._v("in.hist", Bp, s.Z), Bp, s.Z)
        if z is not None:
            ops.copy2d(zh[:, :s.Z], eng._v("in.z", Bp, s.Z), Bp, s.Z)
        if s.add_dim:
            ops.copy2d(zh[:, s.zin - s.add_dim:], eng._v("in.add", Bp, s.add_dim), Bp, s.add_dim)
        eng._have_staged_targets = have_targets
        return B

    def _worker_submit(self, fn):
        """run ``fn`` on the stager's worker thread (started on first use; it has made the engine's device current once)"""
        if self._worker is None:
            self._jobs = queue.Queue()

            def loop():
                torch.cuda.set_device(self.eng.device)
                while True:
                    job = self._jobs.get()
                    if job is None:
                        return
                    job()
            self._worker = threading.Thread(target=loop, name="mvae-stage-prefetch", daemon=True)
            self._worker.start()
        self._jobs.put(fn)

    def prefetch(self, lo, hi, **kw):
        """Start converting minibatch [lo, hi) - both halves - on a worker thread into the mirror the NEXT ``stage`` call uses; that
        call must pass the same arguments and ``prefetched=True``.  For the caller that is about to enqueue a train step: the paced
        host spends most of that call waiting for the device, and the conversion of a 256-window minibatch of float64
        one-hot rows (native, the GIL released) takes 0.5-1 ms per half on a warm 8-thread pool - several ms when the pool is large
        and has slept through the step (profiles/r05_t_pace_mask_by_caller.txt)."""
        kw.pop("prefetched", None)
        return self.stage(lo, hi, _prefetch=True, **kw)

    def finish_targets(self):
        """second half of ``stage(defer_targets=True)``: the decoder heads' targets and row weights, converted now (the encoder is
        already enqueued) and uploaded behind it on the same stream"""
        if self._late is None:
            return
        k, late = self._late
        self._late = None
        if late is not None:
            late()
        cut = self.eng._in_late_off
        self.eng._in_block[cut:].copy_(self.host[k][cut:], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[k] = ev
