"""Step plans: record the C-ABI calls of a step once, replay them with ONE call (include/midivae_hip.h 'STEP PLANS').

What this replaces: the reference enters its backend once per minibatch - ``autoencoder.fit`` runs the Keras train_function
compiled once (reference vae_training.py:804-809; evaluate :300, predict :289,795).  This engine's step is ~70 launches of the C
ABI plus the event / value packets that order them over 7 queues; issued from CPython one by one they take 2.5-3.1 ms per step
(profiles/r03_d_host_vs_device.txt), which is more than the device needs at the reference's shipped shape (T=64).

How: while a ``Recorder`` is active every stream-taking entry point called through ``hiplib`` is executed as usual AND noted
(name, argument values, a copy of every argument struct).  Three recordings of the same kind of step are compared field by field:
whatever is equal in all three is a constant of the plan; a 32-bit field that differs must hold a COUNTER VALUE - the engine
hands those around as ``ops.CounterValue`` (the cumulative progress counters of the time-pipelined kernels, the sequence numbers
of value joins), whose fields the marshalling layer announces (``Recorder.note_field``) - and becomes a patch
``field = counter_before_the_step + offset``.  Anything else that
differs - a pointer that moved, a changed shape - means the step is not replayable and the engine keeps enqueueing from Python.
Replay (``StepPlan.run``) hands the current counter values to ``mvae_plan_run`` and advances them by what one step adds.
Nothing here computes: no oracle, no fallback arithmetic.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import hiplib as hl

# entry points a plan may hold (csrc/plan.cpp k_entries): everything that takes a stream
RECORDABLE = (
    "mvae_rnn_fwd", "mvae_rnn_bwd", "mvae_rnn_fwd_multi", "mvae_rnn_bwd_multi", "mvae_pack_recurrent", "mvae_gemm",
    "mvae_gemm_kstream_multi", "mvae_gemm_multi", "mvae_colsum", "mvae_stream_wait_value32", "mvae_stream_write_value32", "mvae_prepare_batch",
    "mvae_outer_bias_tile16", "mvae_gather2_tile16", "mvae_colsum_weighted", "mvae_sum_over_time", "mvae_head", "mvae_latent_fwd",
    "mvae_latent_bwd", "mvae_latent_chain_fwd", "mvae_latent_chain_bwd", "mvae_relayout", "mvae_tanh_bwd", "mvae_convert",
    "mvae_make_table", "mvae_transpose_convert", "mvae_adam_step", "mvae_adam_step_dev", "mvae_rmsprop_step",
    "mvae_scalars_accumulate", "mvae_copy2d_f32", "mvae_history_from_latent", "mvae_signature_head_fwd",
    "mvae_signature_head_bwd", "mvae_softmax_bwd_add", "mvae_bi_concat", "mvae_add_time_reversed", "mvae_event_record",
    "mvae_stream_wait_event")
_M64 = (1 << 64) - 1
_active = None          # the Recorder noting calls right now (one at a time: the engine's enqueue is single-threaded)


def active():
    return _active


def _encode(argtypes, args):
    """one 64-bit slot per argument (+ the bytes of every struct / array argument)"""
    slots, blobs = [], {}
    for i, (t, a) in enumerate(zip(argtypes, args)):
        if isinstance(a, (C.Structure, C.Array)):
            blobs[i] = C.string_at(C.addressof(a), C.sizeof(a))
            slots.append(0)
        elif a is None:
            slots.append(0)
        elif t is C.c_float:
            slots.append(struct.unpack("<I", struct.pack("<f", float(a)))[0])
        else:
            slots.append(int(a) & _M64)
    return slots, blobs


_VIEW_OPS = ("view", "_unsafe_view", "reshape", "slice", "select", "as_strided", "alias", "detach", "t", "transpose", "permute",
             "expand", "unsqueeze", "squeeze", "narrow", "unbind", "split", "split_with_sizes", "view_as", "lift_fresh", "_reshape_alias",
             "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "is_contiguous", "size", "stride", "dim", "numel")


def _taint_mode(rec):
    """a torch dispatch mode that notes the first torch operation which is more than a view: a kernel torch launches inside a
    recorded step would be missing from the replay, so such a step is never armed"""
    from torch.utils._python_dispatch import TorchDispatchMode

    class _Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = getattr(getattr(func, "overloadpacket", None), "__name__", str(func))
            if name not in _VIEW_OPS and rec.tainted is None:
                rec.tainted = name
            return func(*args, **(kwargs or {}))
    return _Mode()


class Recorder(object):
    """``with Recorder() as rec: <enqueue a step>`` - rec.calls = [(name, slots, {slot: bytes})], rec.tags = {(call, argument,
    byte offset or -1): counter key}, rec.tainted = name of the first torch operation (other than a view) the step ran, or None"""

    def __init__(self):
        self.calls, self.tags, self._saved, self.tainted, self._mode, self._pending = [], {}, {}, None, None, []
        self.marks = []         # host marks: (index of the next call, tag, stream handle current at the mark)

    def host(self, tag, stream_handle, fn):
        """a HOST action in the middle of the step that a replay must repeat at the same place - a data-parallel step's
        ``dist.all_reduce`` calls (torch / RCCL enqueue them, not this library): noted as a mark between two calls, run now with
        the taint mode suspended (what it enqueues is the host action's own business, on replay as now)"""
        self.marks.append((len(self.calls), tag, int(stream_handle)))
        self._mode.__exit__(None, None, None)
        try:
            return fn()
        finally:
            self._mode = _taint_mode(self)
            self._mode.__enter__()

    def note_field(self, arg_index, offset, key):
        """the NEXT library call carries a value of the engine's counter ``key`` in argument ``arg_index`` (at byte ``offset`` of
        the struct(s) it points at; -1: the scalar argument itself) - ops.CounterValue announces itself here"""
        self._pending.append((arg_index, offset, key))

    def __enter__(self):
        global _active
        assert _active is None, "recorders do not nest"
        lib = hl.load()
        for name in RECORDABLE:
            real, argtypes = getattr(lib, name), hl.SIGNATURES[name][1]
            self._saved[name] = real

            def wrapper(*args, _real=real, _name=name, _argtypes=argtypes):
                rc = _real(*args)
                pending, self._pending = self._pending, []
                if rc == 0:          # (a rejected call enqueued nothing: the engine takes another path)
                    slots, blobs = _encode(_argtypes, args)
                    for arg_index, offset, key in pending:
                        self.tags[(len(self.calls), arg_index, offset)] = key
                    self.calls.append((_name, slots, blobs))
                return rc
            setattr(lib, name, wrapper)
        _active = self
        self._mode = _taint_mode(self)
        self._mode.__enter__()
        return self

    def __exit__(self, *exc):
        global _active
        if self._mode is not None:
            self._mode.__exit__(*exc)
            self._mode = None
        lib = hl.load()
        for name, real in self._saved.items():
            setattr(lib, name, real)
        _active = None
        return False


class NotReplayable(Exception):
    pass


class StepPlan(object):
    """the C-side plan of one kind of step + the counters it advances"""

    def __init__(self, recordings, marks=()):
        """``recordings``: three (calls, tags, counters_before, counters_after) of the same kind of step; ``marks``: the host
        marks of each recording (Recorder.marks) - a replay runs the call ranges between them (run_ranges)"""
        assert len(recordings) == 3
        marks = [list(m) for m in marks] or [[], [], []]
        if not (marks[0] == marks[1] == marks[2]):
            raise NotReplayable("the host actions of the three recordings differ: %r / %r / %r" % tuple(marks))
        self.marks = marks[0]
        (ca, ta, pa, qa), (cb, tb, pb, qb), (cc, tc, pc, qc) = recordings
        if not (len(ca) == len(cb) == len(cc)):
            raise NotReplayable("the three recordings hold %d / %d / %d calls" % (len(ca), len(cb), len(cc)))
        keys = sorted(set(pa) | set(qa), key=repr)
        self.inc = {}
        for k in keys:
            if k[0] == "param":          # (a call parameter, named by the caller at every replay: not a counter the step advances)
                continue
            inc = [q.get(k, 0) - p.get(k, 0) for p, q in ((pa, qa), (pb, qb), (pc, qc))]
            if not inc[0] == inc[1] == inc[2]:
                raise NotReplayable("counter %r advanced by %r in the three recordings" % (k, inc))
            if inc[0]:
                self.inc[k] = inc[0]
        # PARAMETERS of the call (ops.ParamInt / ParamFloat, key = ("param", name)): the caller names their values at every replay;
        # every field tagged with one is a patch whether or not the three recordings differ in it
        params = sorted({k for t in (ta, tb, tc) for k in t.values() if k[0] == "param"}, key=repr)
        for k in params:
            if not (k in pa and k in pb and k in pc):
                raise NotReplayable("parameter %r was used by a call whose caller did not name its value" % (k,))
        self.params = params
        self.keys = sorted(set(self.inc) | set(params), key=repr)
        kidx = {k: i for i, k in enumerate(self.keys)}
        lib = hl.load()
        h = C.c_void_p()
        hl.check(lib.mvae_plan_create(C.byref(h)), "mvae_plan_create")
        self._h, self._lib = h, lib
        self.n_patches = 0

        def patch_for(where, field, va, vb, vc):
            """the counter key + offset that explains a field taking the values va / vb / vc in the three recordings"""
            ka, kb, kc = ta.get(field), tb.get(field), tc.get(field)
            if ka is None or ka != kb or ka != kc or ka not in kidx:
                raise NotReplayable("%s differs between recordings (%d / %d / %d) and is not a counter value" % (where, va, vb, vc))
            off = [(v - p.get(ka, 0)) & 0xFFFFFFFF for v, p in ((va, pa), (vb, pb), (vc, pc))]
            if not off[0] == off[1] == off[2]:
                raise NotReplayable("%s: counter %r at offsets %r" % (where, ka, off))
            o = off[0] if off[0] < (1 << 31) else off[0] - (1 << 32)
            return kidx[ka], o

        try:
            for i, ((na, sa, ba), (nb, sb, bb), (nc, sc, bc)) in enumerate(zip(ca, cb, cc)):
                if not (na == nb == nc) or not (len(sa) == len(sb) == len(sc)) or not (set(ba) == set(bb) == set(bc)):
                    raise NotReplayable("call %d is %s / %s / %s" % (i, na, nb, nc))
                arr = (C.c_uint64 * len(sa))(*sa)
                ci = lib.mvae_plan_add_call(h, na.encode(), arr, len(sa))
                if ci != i:
                    raise NotReplayable("mvae_plan_add_call(%s) -> %d" % (na, ci))
                for slot in sorted(ba):
                    if not (len(ba[slot]) == len(bb[slot]) == len(bc[slot])):
                        raise NotReplayable("call %d (%s): argument %d changes size" % (i, na, slot))
                    hl.check(lib.mvae_plan_set_blob(h, i, slot, ba[slot], len(ba[slot])), "mvae_plan_set_blob")
                patches = []
                for j, (va, vb, vc) in enumerate(zip(sa, sb, sc)):
                    if va == vb == vc:
                        continue
                    if max(va, vb, vc) >> 32:
                        raise NotReplayable("call %d (%s): argument %d (a pointer?) differs between recordings" % (i, na, j))
                    patches.append((j, -1) + patch_for("call %d (%s) argument %d" % (i, na, j), (i, j, -1), va, vb, vc))
                for slot in sorted(ba):
                    if ba[slot] == bb[slot] == bc[slot]:
                        continue
                    n4 = len(ba[slot]) // 4
                    wa, wb, wc = (np.frombuffer(b, dtype="<u4", count=n4) for b in (ba[slot], bb[slot], bc[slot]))
                    if ba[slot][n4 * 4:] != bb[slot][n4 * 4:] or ba[slot][n4 * 4:] != bc[slot][n4 * 4:]:
                        raise NotReplayable("call %d (%s): argument %d differs in its tail bytes" % (i, na, slot))
                    for w in np.nonzero((wa != wb) | (wa != wc))[0]:
                        patches.append((slot, int(w) * 4) + patch_for("call %d (%s) argument %d byte %d" % (i, na, slot, int(w) * 4),
                                                                      (i, slot, int(w) * 4), int(wa[w]), int(wb[w]), int(wc[w])))
                done = {(slot, off) for slot, off, _, _ in patches}
                for (ci, slot, off), key in sorted(ta.items(), key=repr):         # parameter fields that happened to agree
                    if ci != i or key[0] != "param" or (slot, off) in done:
                        continue
                    if off < 0:
                        va, vb, vc = sa[slot], sb[slot], sc[slot]
                    else:
                        va, vb, vc = (int(np.frombuffer(b[slot], dtype="<u4", count=1, offset=off)[0]) for b in (ba, bb, bc))
                    patches.append((slot, off) + patch_for("call %d (%s) argument %d byte %d" % (i, na, slot, off), (i, slot, off),
                                                          va, vb, vc))
                for slot, off, key, add in patches:
                    hl.check(lib.mvae_plan_add_patch(h, i, slot, off, key, add), "mvae_plan_add_patch")
                    self.n_patches += 1
        except Exception:
            self.close()
            raise
        self.n_calls = len(ca)
        self._vals = (C.c_uint64 * max(len(self.keys), 1))()

    def run(self, counters, first=0, last=-1):
        """enqueue the step for the counter values ``counters`` (dict key -> value BEFORE the step); returns the counters after it"""
        for i, k in enumerate(self.keys):
            self._vals[i] = counters.get(k, 0) & _M64
        rc = self._lib.mvae_plan_run(self._h, first, last, self._vals, len(self.keys))
        if rc != 0:
            raise RuntimeError("mvae_plan_run: call %d failed: %s" % (self._lib.mvae_plan_failed_call(self._h),
                                                                       hl.ERRORS.get(rc, rc)))
        return {k: counters.get(k, 0) + d for k, d in self.inc.items()}

    def run_ranges(self, counters, host):
        """like run, with the host actions in between: ``host(tag, stream_handle)`` is called at every mark, between the two call
        ranges it was recorded between (every patch takes the counter values from BEFORE the step, whatever the range)"""
        first = 0
        for idx, tag, st in self.marks:
            if idx > first:
                self.run(counters, first, idx)
            host(tag, st)
            first = idx
        if first < self.n_calls or not self.marks:
            return self.run(counters, first, -1)
        return {k: counters.get(k, 0) + d for k, d in self.inc.items()}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mvae_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
