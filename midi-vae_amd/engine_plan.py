"""Engine mixin: stream / event helpers that go through the C ABI (so a step plan can hold them) and the record -> arm ->
replay life cycle of step plans (plan.py; include/midivae_hip.h 'STEP PLANS').

Reference counterpart: one backend call per minibatch - Keras' compiled train / test / predict functions behind
``autoencoder.fit`` (reference vae_training.py:804-809), ``evaluate`` (:300), ``encoder.predict`` (:289,795).
"""
from __future__ import annotations

import time
import ctypes as C
import os

import torch

from . import hiplib as hl
from . import plan as _plan


class _PlanSlot(object):
    """recordings / the armed plan of one kind of call on one engine"""
    __slots__ = ("recs", "plan", "dead", "post", "runs", "host_results", "brackets")

    def __init__(self):
        self.recs, self.plan, self.dead, self.post, self.runs, self.host_results, self.brackets = [], None, None, None, 0, {}, []


class PlannedSteps(object):
    # attributes whose value decides which launches a call makes (part of the plan key) and which a call leaves changed (restored
    # after a replay to what the recorded calls left)
    _PLAN_STATE = ("_count_pending", "_grads_clean", "_dxp0_clean", "_have_targets", "_S_done", "_pipe_used", "pipeline",
                   "multi_stream", "_have_staged_targets", "lean_sync", "value_join", "phase_multi")
    # schedule knobs a caller (a test, an A/B script) may change on a LIVE engine: part of the plan key, so that a change selects
    # other plans instead of replaying the launch list recorded under the old value
    _PLAN_CONFIG = ("head_slices", "grad_portions", "index_dense",
                    "index_dense_blocks", "xpand_blocks", "pipe_chunk", "phase_max_B", "kstream_grads", "kstream_wgs", "kstream_rows", "kstream_max_B", "flush_before_join", "dec_kstream", "hold_side_heads",
                    "kstream_singles", "pipe_gemm_blocks", "pipe_proj_blocks", "time_chunks", "fuse_head_bwd", "fuse_bias_grad",
                    "fused_latent", "gate_side_heads", "_hold_dec_grads", "_diag_no_param_grads", "use_plans", "defer_grads_rows", "defer_early", "defer_early_rows", "defer_split_wgs", "gate_pipe_gemms", "pace_mask", "pace_mask_split", "_pace_mask_now", "pace_early")
    # ... and the spec's floats that reach kernel arguments as immediates of the recorded launches
    _PLAN_SPEC = ("lr", "beta", "prior_mean", "prior_std", "epsilon_std", "w_instr", "w_vel", "w_style", "w_held", "w_next", "w_sig",
                  "w_cnotes", "w_cinstr", "optimizer")

    _EV_MAX = 8192
    _REBASE = 1 << 30          # device counters are re-based (synchronize + zero) before they reach this (engine._sync_region, _join)

    def _plan_init(self):
        self.use_plans = os.environ.get("MVAE_PLANS", "1") == "1"
        self._plans = {}
        self._ev_pool, self._ev_i = [], 0
        self._host_results, self._ext_streams, self._plan_depth = {}, {}, 0
        self._prof_pool, self._prof_pending, self._prof_i, self._prof_call = [], {}, 0, []     # launch brackets (Engine._timed)
        self.pace_wait_s = 0.0      # host seconds spent waiting for the device by design (steps_in_flight, pace_mask): not host WORK
        self.steps_in_flight = int(os.environ.get("MVAE_STEPS_IN_FLIGHT", "1"))      # (_planned; 0 = the host runs ahead freely)
        self._flight, self._flight_pool = [], []
        self.plan_stats = {"recorded": 0, "replayed": 0, "refused": {}}

    # ---- events / stream ordering through the C ABI ---------------------------------------------------------------------------
    def _ev_next(self):
        """the next event of the engine's ring (reused call after call: a wait takes the record that precedes it in program order)"""
        if self._ev_i >= len(self._ev_pool):
            h = C.c_void_p()
            hl.check(hl.load().mvae_event_create(C.byref(h)), "mvae_event_create")
            self._ev_pool.append(h.value)
        ev = self._ev_pool[self._ev_i]
        # The ring restarts with every planned call (_planned: _ev_i = 0); inside ONE call it only grows - a handle re-recorded
        # while an earlier wait on it is still to be enqueued would order that wait against the wrong point.  Callers
        # outside _planned (data-parallel steps run from Python) restart it at a step boundary (_step_begin).
        self._ev_i += 1
        if self._ev_i >= self._EV_MAX:
            raise RuntimeError("one call drew %d events: the schedule is not meant to (time chunks x layers out of hand?)" % self._ev_i)
        return ev

    def _ev_record(self, stream):
        ev = self._ev_next()
        hl.check(hl.load().mvae_event_record(ev, stream.cuda_stream), "mvae_event_record")
        return ev

    def _ev_wait(self, stream, ev):
        hl.check(hl.load().mvae_stream_wait_event(stream.cuda_stream, ev), "mvae_stream_wait_event")

    def _wait_stream(self, waiter, waited):
        """``waiter`` waits for everything enqueued so far on ``waited`` (torch's Stream.wait_stream through the C ABI)"""
        self._ev_wait(waiter, self._ev_record(waited))

    # ---- plans ------------------------------------------------------------------------------------------------------------
    def _plan_counters(self):
        d = {("sync", slot, i): v for slot, cum in self._sync_cum.items() for i, v in enumerate(cum)}
        d.update({("join", w): v for w, v in self._join_seq.items()})
        return d

    def _plan_set_counters(self, d):
        for k, v in d.items():
            if k[0] == "sync":
                self._sync_cum.setdefault(k[1], [0, 0])[k[2]] = v
            else:
                self._join_seq[k[1]] = v

    def _plan_state(self):
        spec = self.spec
        return (tuple(getattr(self, n, None) for n in self._PLAN_STATE) +
                tuple(getattr(self, n, None) for n in self._PLAN_CONFIG) +
                tuple(getattr(spec, n, None) for n in self._PLAN_SPEC) +
                (frozenset(self._xp0_bias), tuple(sorted(self.start_zero.items())), bool(self._weights_dirty),
                 frozenset(self._pipe_verified), self._prep is None, getattr(self, "status_allreduce", None) is not None))

    def _plan_post(self, before):
        """what a call left changed, as (attribute values, weight-version moves) relative to the state ``before`` it"""
        return (tuple(getattr(self, n, None) for n in self._PLAN_STATE), frozenset(self._xp0_bias),
                self._pver[0] - before[0], self._prepared_ver - self._pver[0])

    def _prof_harvest(self, i):
        """read the measurement bracket pair ``i`` still holds (it is about to be recorded again, or the summary is due)"""
        pend = self._prof_pending.pop(i, None)
        if pend is not None and self.prof is not None:
            ms = C.c_float()
            e0, e1 = self._prof_pool[i]
            hl.check(hl.load().mvae_event_elapsed_ms(e0, e1, C.byref(ms)), "mvae_event_elapsed_ms")
            self.prof.setdefault(pend[0], []).append((float(ms.value), pend[1]))

    def _host_call(self, tag, fn):
        """a host action inside a step (a data-parallel step's collectives): run ``fn`` - noted as a host mark when the step is
        being recorded, so that a replay of the step runs the table entry ``tag`` of its caller at the same place"""
        rec = _plan.active()
        if rec is None:
            return fn()
        out = rec.host(tag, torch.cuda.current_stream().cuda_stream, fn)
        self._host_results[tag] = out
        return out

    def _replay_host(self, host, want):
        """the callback StepPlan.run_ranges hands a mark to: run the caller's host action ``tag`` with the stream current that was
        current when it was recorded.  The critical stream is simply still current (a replay runs where its recording ran: the
        stream is part of the plan key); any other one is one of the engine's own torch streams (the communication stream of the
        early bucket).  NOT torch.cuda.ExternalStream(handle): wrapped that way the default stream (handle 0) is another stream to
        torch - a collective issued under it was not ordered against the launches around it (measured: two-rank fit 3e-4 off)."""
        def run(tag, handle):
            cur = torch.cuda.current_stream()
            if handle == cur.cuda_stream:
                out = host[tag]()
            else:
                st = self._ext_streams.get(handle)
                if st is None:
                    own = [self.s_comm, self.s_grad, self.s_grad2, self.s_vel, self.s_instr, self.s_held, self.s_next, *self.s_layer, *self.s_proj]
                    st = next((x for x in own if x is not None and x.cuda_stream == handle), None)
                    if st is None:
                        raise RuntimeError("host action %r was recorded on a stream the engine does not own" % tag)
                    self._ext_streams[handle] = st
                with torch.cuda.stream(st):
                    out = host[tag]()
            if out != want.get(tag):         # (e.g. the gradient scale a hook returns: a constant of the recorded optimizer launch)
                raise RuntimeError("host action %r returned %r; the recorded step was built for %r" % (tag, out, want.get(tag)))
        return run

    # kinds of calls that START a unit of work the host should not run ahead of (steps_in_flight), and that END one
    # (train steps only: decoding BASELINE configs[4] measured 11.07 ms per call unpaced, 11.14 paced)
    _PACE_START = ("train", "train_begin", "train_begin_fused")
    _PACE_END = ("train", "train_finish")

    def _planned(self, kind, fn, host=None, params=None):
        """_planned_call, paced: with ``steps_in_flight`` = n > 0 the host enqueues a step only when at most n - 1 earlier ones are
        still unfinished (it waits for the event behind the n-th last).  Round 5 measurement: a device that has the NEXT step's
        packets in its queues - a dozen hardware queues holding value waits that will not be satisfied for milliseconds - runs the
        CURRENT step slower than one whose host enqueues step i + 1 only after step i has ended, although the enqueue (0.2 ms of
        plan replay) is then exposed: reference shape T = 64 2.05 -> 1.63-1.68 ms per step (GRU and LSTM), BASELINE configs[1]
        6.73 -> 6.56 (LSTM), 5.67 -> 5.59 (GRU); two steps in flight already lose all of it (profiles/r05_c_steps_in_flight.txt)."""
        outer = self._plan_depth == 0 and _plan.active() is None
        n = self.steps_in_flight if outer else 0
        if n and kind[0] in self._PACE_START:
            while len(self._flight) >= n:
                ev = self._flight.pop(0)
                t0 = time.perf_counter()
                ev.synchronize()
                self.pace_wait_s += time.perf_counter() - t0
                self._flight_pool.append(ev)
        try:
            return self._planned_call(kind, fn, host, params)
        finally:
            if n and kind[0] in self._PACE_END:
                ev = self._flight_pool.pop() if self._flight_pool else torch.cuda.Event()
                ev.record()
                self._flight.append(ev)
                if len(self._flight) > 8:                      # (an unpaced caller in between: nothing to wait for)
                    self._flight_pool.append(self._flight.pop(0))

    def _planned_call(self, kind, fn, host=None, params=None):
        """run ``fn`` (the Python enqueue of one call of kind ``kind``, a hashable that names everything the launch list depends
        on besides the engine's state) - or, once three recordings of it agreed, replay its plan.  ``host``: tag -> callable of
        the host actions ``fn`` performs through _host_call (replayed between the call ranges they were recorded between)"""
        profiling = self.prof is not None and (self.prof_kinds is None or len(self.prof_kinds) > 0)      # (launches get bracketed)
        if self._plan_depth == 0:
            self._ev_i = 0          # the event ring restarts with every outermost call (planned or not)
            self._prof_i, self._prof_call = 0, []       # ... and so do the launch brackets
        if (not self.use_plans or getattr(self, "marks", None) is not None or
                _plan.active() is not None or self._hist_fused is not None):
            self._plan_depth += 1
            try:
                return fn()
            finally:
                self._plan_depth -= 1
        pkey = (None if not profiling else True if self.prof_kinds is None else frozenset(self.prof_kinds))      # which launches are bracketed
        key = (kind, torch.cuda.current_stream().cuda_stream, self._plan_state(), pkey)
        slot = self._plans.get(key)
        if slot is None:
            slot = self._plans[key] = _PlanSlot()
        if slot.plan is not None:
            # (a counter about to wrap is re-based by the Python path - _sync_region / _join, both at 2^30 - which then runs WITHOUT a
            #  recorder: the armed plan stays, and replays again from the next call on)
            cnt = self._plan_counters()
            if not all(cnt.get(k, 0) + d < self._REBASE for k, d in slot.plan.inc.items()):
                return fn()
            if params:
                cnt.update(params)          # (the call's parameters: the real number of windows, 1 / global minibatch size)
            if True:
                for i, _, _ in slot.brackets:            # (their pairs are recorded again by this replay)
                    self._prof_harvest(i)
                if slot.plan.marks:
                    self._plan_set_counters(slot.plan.run_ranges(cnt, self._replay_host(host or {}, slot.host_results)))
                else:
                    self._plan_set_counters(slot.plan.run(cnt))
                vals, xp0, dver, dprep = slot.post
                for n, v in zip(self._PLAN_STATE, vals):
                    setattr(self, n, v)
                self._xp0_bias = set(xp0)
                self._pver[0] += dver
                self._prepared_ver = self._pver[0] + dprep
                for i, k, n in slot.brackets:
                    self._prof_pending[i] = (k, n)
                slot.runs += 1
                self.plan_stats["replayed"] += 1
                return None
        if slot.dead is not None:
            return fn()
        pre, ver = self._plan_counters(), (self._pver[0], self._prepared_ver)
        if params:
            pre.update(params)
        self._host_results = {}
        with _plan.Recorder() as rec:
            out = fn()
        post_state = self._plan_post(ver) + (tuple(sorted(self._host_results.items(), key=repr)),)
        slot.recs.append((rec.calls, rec.tags, pre, self._plan_counters(), post_state, rec.marks))
        slot.brackets = list(self._prof_call)
        self.plan_stats["recorded"] += 1
        if rec.tainted is not None:      # (a kernel torch itself launched would be missing from the replay)
            slot.dead = "the call runs the torch operation %r" % rec.tainted
            self.plan_stats["refused"][repr(kind)] = slot.dead
            slot.recs = []
        elif len(slot.recs) >= 3:
            a, b, c = slot.recs[-3:]
            if not (a[4] == b[4] == c[4]):
                slot.dead = "the call leaves the engine in different states"
            else:
                try:
                    slot.plan, slot.post = _plan.StepPlan([r[:4] for r in (a, b, c)], [r[5] for r in (a, b, c)]), c[4][:4]
                    slot.host_results = dict(c[4][4])
                except _plan.NotReplayable as e:
                    slot.dead = str(e)
            if slot.dead is not None:
                self.plan_stats["refused"][repr(kind)] = slot.dead
            slot.recs = []
        return out

    def drop_plans(self):
        """forget every plan (buffers re-allocated, streams changed, ...)"""
        for s in self._plans.values():
            if s.plan is not None:
                s.plan.close()
        self._plans = {}
