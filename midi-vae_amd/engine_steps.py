"""Engine mixin: the calls a caller makes per minibatch - train_step / train_step_begin + finish (reference: ONE Keras
train_function call per minibatch, vae_training.py:804-809), their replay as plans, the paced host and the gradient hooks of data
parallelism."""
from __future__ import annotations

import time
import ctypes as C

import torch

from . import hiplib as hl
from . import ops


class TrainSteps(object):
    def _step_begin(self):
        """weight preparation (or, with unchanged weights, the zeroing it would have done) and clean gradient buffers"""
        assert self.training
        self._have_targets = True
        if self._weights_dirty:
            self.prepare_weights()          # (also zeroes the loss / metric accumulators and the constant-input cells' dxp0 sums)
            self._dxp0_clean = True
        else:
            self._zero_scal()
            self._dxp0_clean = False
        if not self._grads_clean:
            self._zero_grads()
        self._grads_clean = False

    def _zero_grads(self):
        """the gradient buffer of a step that does not follow an optimizer step (which leaves it zeroed): a prepare-batch job like
        _zero_scal, so that this state of a step replays as a plan too"""
        if self._zero_grads_job is None:
            self._zero_grads_job = ops.PrepBatch()
            self._zero_grads_job.zero(self.grads)
        self._zero_grads_job.run()

    def _zero_scal(self):
        """the loss / metric accumulators of a call that does not prepare weights (a one-job mvae_prepare_batch: part of a step
        plan, which a torch fill would not be)"""
        if self._zero_scal_job is None:
            self._zero_scal_job = ops.PrepBatch()
            self._zero_scal_job.zero(self.scal)
        self._zero_scal_job.run()

    def _redo_step(self, B):
        """the forward + backward pass of a train step once more (first use of the pipelined kernels stalled: _verify_pipeline).
        A fused history pre-pass is redone with it (the history rows came out of the timed-out forward); the gradient hook is
        out of the way (_overlap_hook: no early bucket is in flight on an unverified step), so the buffer may be zeroed"""
        self.scal.zero_()
        self.grads.zero_()
        self._hist_fused = self._redo_hist
        try:
            self.encoder_forward(B, with_init=True)
        finally:
            self._hist_fused = None
        self.decoder_forward(B)
        self.backward(B)

    def forward_backward(self, B):
        """One pass of forward + losses + backward on the staged batch (gradients left in self.grads)."""
        self._mark("step start")
        self._step_begin()
        self._mark("weights prepared")
        self.encoder_forward(B, with_init=True)
        self._mark("encoder forward (incl. latent)")
        # The velocity / instrument branches' backward depends on nothing the notes branch does in between: no join at
        # the end of the decoder forward pass and no fork at the start of the backward pass (two packets less on the
        # critical queue); they are joined where the decoder BPTT ends.
        self._branches_stay_forked = self.lean_sync and self.multi_stream and not self.aux
        try:
            self._pace(1)
            self.decoder_forward(B)
            self._mark("decoder forward + heads")
            self._pace(2)
            self.backward(B)
        finally:
            self._branches_stay_forked = False
        self._mark("backward")
        self._verify_pipeline(lambda: self._redo_step(B))

    def _verify_pipeline(self, redo, key="train"):
        """First use of time-pipelined stacks by each kind of call (train step / encode / decode / predict): make sure no kernel
        gave up waiting for its producer.  A first use can stall for seconds for reasons that do not repeat - first launches of the
        kind's kernels (code-object loads, hipFuncSetAttribute) and first pinned / device allocations of the caller's staging beside
        it, all of which hold new dispatches back while a WAITING kernel is resident - so the work is first redone as it is; if a
        kernel gives up again (two of the engine's streams share a hardware queue, or kernels run one at a time under counter
        collection) the engine falls back to one launch per chunk for good and redoes it once more."""
        if not self.pipeline or key in self._pipe_verified or not self._pipe_used:
            return                     # (a call whose batch did not run any stack pipelined verifies nothing)
        self._pipe_verified.add(key)

        def status():
            # (data parallel: the decision stays rank-local - a rank whose shard is empty never gets here, so a collective in
            #  this place could hang - and that is safe: a redo issues no collective (no early bucket on an unverified step,
            #  _overlap_hook), recomputes the same gradients, and the per-step status all-reduce of optimizer_step is issued by
            #  every rank whatever schedule it ended up with)
            return int(self.store["pipe_status"].item())
        if status() == 0:
            return
        self.store["pipe_status"].zero_()
        self._dxp0_clean = False
        redo()
        if status() != 0:
            import warnings
            warnings.warn("time-pipelined recurrent kernels timed out waiting for their producers (status %d; call %r); falling "
                          "back to chunked launches (Engine.pipeline = False)" % (int(self.store["pipe_status"].item()), key))
            self.store["pipe_status"].zero_()
            self.pipeline = False
            self._dxp0_clean = False
            redo()

    def train_step_begin(self, B, hist_fused=None):
        """First part of a train step - weight preparation and the encoder up to the sampled z - for callers that stage the
        decoder heads' targets while it runs (Stager.stage(defer_targets=True) ... Stager.finish_targets()); the rest:
        train_step_finish.

        ``hist_fused`` = (eps2, z_out): the FUSED HISTORY PRE-PASS (reference vae_training.py:788-798 + :804-809 in one encoder
        forward).  The reference runs ``encoder.predict`` over the song - with the weights this step starts from and a fresh
        draw eps2 - to obtain the history input H[i] = z'[i-1] of ``fit``, then the step's own encoder forward with another
        draw.  Same weights, same inputs, same mu / log sigma^2: here z' = mu + sigma * eps2 comes out of THIS step's encoder
        forward (eps2: (Bp, Z) device view, already scaled; z_out: (>= B, Z) device rows that receive z'), is rolled into the
        history columns of [z | history] (window 0: zeros) and the decoder's initial-state Denses follow as a separate GEMM.
        Only for a minibatch that starts at window 0 of its song."""
        if hist_fused is None:
            return self._planned(("train_begin",) + self._kind_B(B), lambda: self._train_step_begin(B, None), params=self._call_params(B))
        # z' goes to a fixed engine buffer - the launch list then holds no per-song address and the step replays as a plan like
        # any other - and to the caller's rows by one copy behind the step (train_step_finish)
        eps2, z_dst = hist_fused
        zbuf = self._v("hist_zout", self.pad16(B), self.spec.Z)
        self._fused_dst = (z_dst, zbuf)
        return self._planned(("train_begin_fused",) + self._kind_B(B) + (eps2.data_ptr(),), lambda: self._train_step_begin(B, (eps2, zbuf)),
                             params=self._call_params(B))

    def _train_step_begin(self, B, hist_fused):
        self._step_begin()
        self._hist_fused = self._redo_hist = hist_fused
        try:
            self.encoder_forward(B, with_init=True)
        finally:
            self._hist_fused = None

    def train_step_finish(self, B, allreduce=None):
        """the rest of the step: one replayable call (engine_plan.py) - with a gradient hook (data parallel) its collectives are
        host actions between the call ranges of the plan"""
        self._pace_mask_now = self.pace_mask_split
        try:
            return self._planned(("train_finish",) + self._kind_B(B) + self._hook_kind(allreduce),
                                 lambda: self._train_step_finish(B, allreduce), host=self._hook_table(allreduce), params=self._call_params(B))
        finally:
            self._pace_mask_now = self.pace_mask
            if self._fused_dst is not None:          # (behind a possible redo of the step: the rows are final here)
                (z_dst, zbuf), self._fused_dst = self._fused_dst, None
                z_dst.copy_(zbuf[:z_dst.shape[0]])

    def _overlap_hook(self, allreduce, B):
        """the hook whose decoder bucket is reduced beside the encoder BPTT - not on a step that may still be redone (the first
        pipelined train step of an engine, _verify_pipeline): the redo zeroes and recomputes the gradients, which must not race a
        collective already in flight on part of them; that one step reduces the whole buffer afterwards"""
        unverified = self.pipeline and "train" not in self._pipe_verified
        # (... nor on a step that defers its weight-gradient GEMMs: the decoder bucket is complete only behind the last recurrence)
        return allreduce if (getattr(allreduce, "overlap", False) and not unverified and not self._defers_grads(self.pad16(B))) else None

    def _train_step_finish(self, B, allreduce):
        self._bucket_hook = self._overlap_hook(allreduce, B)
        self._branches_stay_forked = self.lean_sync and self.multi_stream and not self.aux
        try:
            self.decoder_forward(B)
            self._pace(2)
            self.backward(B)
        finally:
            self._branches_stay_forked = False
            self._bucket_hook = None
        self._verify_pipeline(lambda: self._redo_step(B))
        gs = self._host_call("reduce", lambda: allreduce(self.grads)) if allreduce is not None else 1.0
        self.optimizer_step(gs if gs is not None else 1.0)

    def train_step(self, B, allreduce=None):
        """forward + backward + (optional gradient all-reduce hook) + optimizer update on the staged batch.  The whole step is one
        replayable call: after three recorded steps its ~70 launches are enqueued by mvae_plan_run (engine_plan.py; reference: ONE
        Keras train_function call per minibatch, vae_training.py:804-809); the hook's collectives are issued from Python between
        the plan's call ranges (host marks)."""
        self._pace_mask_now = self.pace_mask
        return self._planned(("train",) + self._kind_B(B) + self._hook_kind(allreduce), lambda: self._train_step(B, allreduce),
                             host=self._hook_table(allreduce), params=self._call_params(B))

    def _call_params(self, B):
        """the parameters of a call on B windows (ops.PARAM_*): what a replayed plan patches into the launches that take them"""
        return {ops.PARAM_B: int(B), ops.PARAM_INV_BATCH: ops.f32_bits(1.0 / self.norm_B)}

    def _kind_B(self, B):
        """what the number of windows contributes to the plan key of a train step.  On the default graph (fused latent chain, no
        optional heads) a step's launch list depends on the PADDED batch only - the real count and the loss normaliser reach the
        latent chain and the heads as tagged parameters (ops.ParamInt / ParamFloat) - so songs of 97 and 100 windows share a
        plan: `python vae_training.py` on songs of 20-200 windows replays after three steps per 16-window bucket instead of three
        per window count.  Any other graph: the real count and the normaliser are part of the key."""
        s = self.spec
        if self.fused_latent and self._chain_ok() and not self._chain_refused and not s.signature and not self.aux:
            return ("padded", self.pad16(B))
        return (int(B), float(self.norm_B))

    def _hook_kind(self, allreduce):
        """what a gradient hook adds to the plan key of a train step: that there is one, and whether it takes an early bucket"""
        return () if allreduce is None else ("hook", bool(getattr(allreduce, "overlap", False)))

    def _hook_table(self, allreduce):
        """the host actions of a data-parallel step by tag (engine_plan._host_call): what a replayed step calls between its ranges
        of launches - Python then issues nothing but the collectives (reference: one train_function call per minibatch)"""
        L = self.layout
        return {"early": lambda: allreduce.early(self.grads[L.dec_begin:L.total]),
                "reduce": lambda: allreduce(self.grads),
                "status": lambda: self.status_allreduce(self.store["pipe_status"]),
                "pace1": lambda: self._pace_now(1), "pace2": lambda: self._pace_now(2), "pace4": lambda: self._pace_now(4)}

    def _pace(self, bit):
        """hold the HOST here until the device has reached the pause's point of the step (pace_mask: bit 1 before the decoder forward,
        2 before the backward pass, 4 before the encoder BPTT) - a host action of the step, so a replayed step pauses there too.
        The point is an event of the pause's own: recorded further up the queue by _pace_point (pace_early), else here."""
        if self._pace_mask_now & bit:
            if bit not in self._pace_recorded:
                self._pace_record(bit)
            self._pace_recorded.discard(bit)
            self._host_call("pace%d" % bit, lambda: self._pace_now(bit))

    def _pace_record(self, bit):
        if bit not in self._pace_events:
            h = C.c_void_p()
            hl.check(hl.load().mvae_event_create(C.byref(h)), "mvae_event_create")
            self._pace_events[bit] = h.value
        hl.check(hl.load().mvae_event_record(self._pace_events[bit], torch.cuda.current_stream().cuda_stream), "mvae_event_record")
        self._pace_recorded.add(bit)

    def _pace_point(self, bit):
        """the point of the step the host pause ``bit`` waits for, when it lies BEFORE the pause (pace_early): the pause further down
        the same queue then ends when the device gets HERE, and the host's wake-up and the first launches of the next call range
        hide behind the kernels in between (the notes head in front of the backward pass, the latent chain in front of the encoder
        BPTT: 35-40 us of idle critical queue each, timeline r05_p)"""
        if self._pace_mask_now & bit and self.pace_early:
            self._pace_record(bit)

    def _pace_now(self, bit):
        t0 = time.perf_counter()
        hl.check(hl.load().mvae_event_synchronize(self._pace_events[bit]), "mvae_event_synchronize")
        self.pace_wait_s += time.perf_counter() - t0          # (bench.py: host time inside a step that is waiting, not work)

    def _train_step(self, B, allreduce):
        self._redo_hist = None
        self._bucket_hook = self._overlap_hook(allreduce, B)
        try:
            self.forward_backward(B)
        finally:
            self._bucket_hook = None
        gs = 1.0
        if allreduce is not None:
            gs = self._host_call("reduce", lambda: allreduce(self.grads))
        self.optimizer_step(gs if gs is not None else 1.0)

    def train_step_empty(self, allreduce):
        """Data parallel, ragged minibatch: this rank's shard is EMPTY (fewer windows than ranks) - contribute zero gradients to
        the collective and apply the same update as everybody else."""
        assert self.training and allreduce is not None
        if self._weights_dirty:
            self.prepare_weights()
        else:
            self.scal.zero_()
        if not self._grads_clean:
            self.grads.zero_()
        self._grads_clean = False
        gs = allreduce(self.grads)
        self.optimizer_step(gs if gs is not None else 1.0)

