"""PHASE launches (mixin of engine.Engine): the recurrences of one phase of the step as ONE launch on the critical queue.

A train step is four serial recurrence phases - encoder forward, decoder forward, decoder BPTT, encoder BPTT (reference
vae_definition.py:443-480, 519-726 and their gradients).  Each phase used to be 3-5 launches on as many HIP streams, forked from
and joined into the critical stream by events; the kernel timeline (profiles/r03_b_timeline_lstm_step.txt) shows what that
costs: a cross-queue dependency takes the command processors 40-120 us to resolve when it resolves late, and every phase boundary
had one or two (~0.5 ms of a 7.7 ms step).  Here every recurrence of an ENCODER phase (the notes stack's layers and the
instrument / velocity / held-notes rolls beside it) and the notes stack of a DECODER phase are problems of one
``mvae_rnn_fwd_multi`` / ``mvae_rnn_bwd_multi`` launch on the critical queue; what follows a phase follows it in queue order
(~2-8 us).  The time-pipelined hand-over inside a stack is what it was (device-side counters; the persistent projection / dX
GEMM between two layers on its own queue, forked by one event and never joined: see Engine.backward); the x*W + b
expansion of the 1-feature velocity roll is a chunk-publishing producer INSIDE the encoder-forward launch.  The decoder's
velocity / instrument heads keep their own queues: they leave the critical queue with one event after the latent chain and are
waited for where the decoder BPTT ends - long after they finished.
Shapes the slot-interleaved kernels do not take (f32 parity mode, H != 256, the bidirectional encoder), stacks whose kernels
cannot all be resident, and Engine.phase_multi = False run the per-stream schedule of engine.py.
"""
from __future__ import annotations

import torch

from . import hiplib as hl
from . import ops


class PhaseLaunches(object):
    def _phase_ok(self, stack, singles):
        """may ``stack`` (bottom -> top) and the single-layer recurrences ``singles`` run as one launch?"""
        recs = list(stack) + list(singles)
        return (self.phase_multi and self._cur_B <= self.phase_max_B and self.tile16 and self.multi_stream and 0 < len(recs) <= 8 and
                all(self._il(r) for r in recs) and (len(stack) < 2 or self._pipelined(stack)) and
                all(r.xmode != hl.X_SCALAR or self._scalar_as_dense(r) for r in recs))

    # ---- forward ----------------------------------------------------------------------------------------------------
    def _xpand_problem(self, r, B, xs=None, idx=None):
        """the x*W + b expansion of a 1-feature roll - or the table rows of a one-hot layer (``idx``) - as a producer inside the
        launch: (xpand args, pipe fields of its consumer)"""
        s, P, p = self.spec, self.P, r.prefix
        cs = self.pipe_chunk
        while r.T % cs:
            cs //= 2
        blocks = self.xpand_blocks if idx is None else self.index_dense_blocks
        sync, target, _ = self._sync_region(5 if idx is None else 9, 1, r.T // cs, self._rnn_waves(r) * blocks, 0)      # (the producer's blocks are blocks of the launch: as many waves each)
        xp = self._v(p + ".xp", r.T, B, s.GH)
        if idx is None:
            x = ops.xpand(xs, P[p + ".W"].view(-1), P[p + ".b"], xp, r.T * B, s.GH, cs * B, sync[0, 0], blocks)
        else:
            x = ops.xpand(None, None, None, xp, r.T * B, s.GH, cs * B, sync[0, 0], blocks, idx=idx,
                          table=self._v(p + ".table", r.K, s.GH))
        return x, dict(chunk_steps=cs, status=self.store["pipe_status"], wait_ready=sync[0, 0], wait_value=target)

    def _stack_problems_forward(self, layers, B, slot, *, states=None, h_last=None, h_last_ld=0, idx=None, start=None, bottom=None):
        """problems of a (pipelined) stack, bottom layer first, and the projection GEMM launches that go with them; ``bottom``:
        pipe fields of a bottom layer whose input projection a producer of the launch writes out (_xpand_problem)"""
        L = len(layers)
        if L == 1:
            r = layers[0]
            return [self._rec_forward(r, B, idx=idx, start=start, h_last=h_last, h_last_ld=h_last_ld, build=True,
                                      pipe=bottom, xp_external=bottom is not None, **(states(r) if states else {}))], []
        cs, T = self.pipe_chunk, layers[0].T
        nchp, nwaves, pwaves = T // cs, self._rnn_waves(layers[0]) * (B // 16), 4 * self.pipe_proj_blocks
        sync, hs_target, xp_target = self._sync_region(slot, L - 1, nchp, nwaves, pwaves)
        self._last_stack_gate = (sync[0, 0][0:1], hs_target)       # the bottom layer's first published chunk of THIS call
        status = self.store["pipe_status"]
        self._pipe_used = True
        probs, gemms = [], []
        for li, r in enumerate(layers):
            top = li == L - 1
            pipe = dict(chunk_steps=cs, status=status)
            if li > 0:
                pipe.update(wait_ready=sync[li - 1, 1], wait_value=xp_target)
            elif bottom is not None:
                assert bottom["chunk_steps"] == cs
                pipe.update(bottom)
            if not top:
                pipe["signal_done"] = sync[li, 0]
            probs.append(self._rec_forward(r, B, idx=idx, start=start, h_last=h_last if top else None,
                                           h_last_ld=h_last_ld if top else 0, pipe=pipe, xp_external=li > 0 or bottom is not None,
                                           build=True,
                                           **(states(r) if states else {})))
            if not top:
                gemms.append((li, lambda li=li: self._rec_xp(
                    layers[li + 1], B, 0, 1, max_blocks=self.pipe_proj_blocks, chunk_rows=cs * B, chunk_wait=sync[li, 0],
                    chunk_wait_value=hs_target, chunk_done=sync[li, 1], chunk_status=status), (sync[li, 0][0:1], hs_target)))
        return probs, gemms

    def _launch_phase_forward(self, key, probs, gemms, xpands=(), steps=0):
        """the phase's launch on this queue, the persistent GEMMs between its layers on their queues - with NO event between the two:
        a GEMM reads nothing before its producer (a problem of the launch, behind the weight preparation on this queue) has
        published a chunk, its queue orders it against the GEMMs of the other phases, and an event record would be one more packet
        (~30-50 us) on the critical queue per phase.  A GEMM that starts early polls."""
        streams = [self.s_proj[li] for li, _, _ in gemms]
        ok = [True]
        self._timed(key, lambda: ok.__setitem__(0, ops.rnn_fwd_multi(probs, xpands)), steps=steps)
        assert ok[0], "mvae_rnn_fwd_multi refused a problem _phase_ok admitted"
        self._launch_pipe_gemms(gemms, streams)

    def _launch_pipe_gemms(self, gemms, streams):
        """the persistent GEMMs between the layers of a phase launch, each on its queue.  (``gate``: a value wait for the first chunk
        its producer publishes, so that the GEMM is not dispatched - resident and polling - a phase early; measured in round 5, no
        gain: T = 64 2.045 -> 2.057 ms, T = 512 6.73 -> 6.81, profiles/r05_c_*; kept as an option for that measurement only)"""
        for (li, fn, gate), st in zip(gemms, streams):
            if self.gate_pipe_gemms:
                ops.stream_wait_value32(gate[0], gate[1], stream=st)
            with torch.cuda.stream(st):
                fn()

    def _encoder_forward_multi(self, B, cat, ldc):
        """every encoder recurrence (reference vae_definition.py:443-480) as one launch; False: not this schedule's"""
        s = self.spec
        H = s.H
        singles = [r for r, _, _ in self.enc_meta]
        if self.enc_bi or not self._phase_ok(self.enc_notes, singles):
            return False
        probs, xpands = [], []
        for k, (r, _, src) in enumerate(self.enc_meta, 1):          # producers and short branches first
            kw = dict(h_last=cat[:, k * H:(k + 1) * H], h_last_ld=ldc, build=True)
            if r.xmode == hl.X_SCALAR:
                x, pipe = self._xpand_problem(r, B, self._v(src, r.T * B))
                xpands.append(x)
                probs.append(self._rec_forward(r, B, pipe=pipe, xp_external=True, **kw))
            else:
                probs.append(self._rec_forward(r, B, idx=self._v(src, r.T, B), **kw))
        x_idx, bottom = self._v("in.x_idx", s.T, B), None
        if self._index_as_dense(self.enc_notes[0]):
            x, bottom = self._xpand_problem(self.enc_notes[0], B, idx=x_idx)
            xpands.insert(0, x)
        sp, gemms = self._stack_problems_forward(self.enc_notes, B, 0, idx=x_idx, h_last=cat[:, 0:H], h_last_ld=ldc, bottom=bottom)
        self._launch_phase_forward(("rnn_fwd_multi", "enc"), sp + probs, gemms, xpands, steps=s.T * len(self.enc_notes))
        return True

    def _notes_forward_multi(self, h, B, states, start):
        """the decoder notes stack (reference vae_definition.py:519-547) as one launch on the critical queue"""
        if len(h.layers) < 2 or not self._phase_ok(h.layers, ()):
            return False
        sp, gemms = self._stack_problems_forward(h.layers, B, 1, states=states, start=start)
        self._launch_phase_forward(("rnn_fwd_multi", "dec"), sp, gemms, steps=h.T * len(h.layers))
        return True

    # ---- backward ---------------------------------------------------------------------------------------------------
    def _stack_problems_backward(self, layers, B, slot, *, dhs_ext=None, dh_last=None, dh_last_ld=0, dstates=None, kstream=False):
        """BPTT problems of a stack, TOP layer first (the producer), and the dX GEMM launches between the layers"""
        order = list(reversed(layers))
        L = len(order)
        if L == 1:
            r = order[0]
            return [self._rec_bptt(r, B, dhs_ext=dhs_ext, dh_last=dh_last, dh_last_ld=dh_last_ld, build=True,
                                   **(dstates(r) if dstates else {}))], [], None
        cs, T = self.pipe_chunk, layers[0].T
        nchp, nwaves, pwaves = T // cs, self._rnn_waves(layers[0]) * (B // 16), 4 * self.pipe_gemm_blocks
        sync, da_target, dx_target = self._sync_region(slot, L, nchp, nwaves, pwaves)
        status = self.store["pipe_status"]
        self._pipe_used = True
        probs, gemms = [], []
        for li, r in enumerate(order):
            top = li == 0
            pipe = dict(chunk_steps=cs, status=status, signal_done=sync[li, 0])       # (every layer publishes: _stack_backward_pipe)
            if li > 0:
                pipe.update(wait_ready=sync[li - 1, 1], wait_value=dx_target)
            ext = dhs_ext if top else self._v(order[li - 1].prefix + ".dx", r.T, B, self.spec.H)
            probs.append(self._rec_bptt(r, B, dhs_ext=ext, dh_last=dh_last if top else None, dh_last_ld=dh_last_ld if top else 0,
                                        pipe=pipe, build=True, **(dstates(r) if dstates else {})))
            if li < L - 1:
                gemms.append((li, lambda li=li, r=r: self._rec_dx(
                    r, B, 0, 1, max_blocks=self.pipe_gemm_blocks, chunk_rows=cs * B, chunk_reverse=True, chunk_wait=sync[li, 0],
                    chunk_wait_value=da_target, chunk_done=sync[li, 1], chunk_status=status), (sync[li, 0][nchp - 1:nchp], da_target)))
        return probs, gemms, (sync, da_target, nchp)

    def _launch_phase_backward(self, key, probs, gemms, steps=0):
        streams = [self.s_proj[li] for li, _, _ in gemms]
        ok = [True]
        self._timed(key, lambda: ok.__setitem__(0, ops.rnn_bwd_multi(probs)), steps=steps)
        assert ok[0], "mvae_rnn_bwd_multi refused a problem _phase_ok admitted"
        self._launch_pipe_gemms(gemms, streams)

    def _notes_backward_multi(self, h, B, dext, dstates, start, head_grads):
        """BPTT through the decoder notes stack as one launch; the output Dense's parameter gradients are released by the launch's
        FIRST published chunk (it runs behind the head kernel on this queue), a layer's by ITS last chunk of da (time steps 0 ..),
        not by the end of the launch - all through device counters, no event on the critical queue"""
        probs, gemms, (sync, da_target, nchp) = self._stack_problems_backward(h.layers, B, 2, dhs_ext=dext, dstates=dstates)
        self._launch_phase_backward(("rnn_bwd_multi", "dec"), probs, gemms, steps=h.T * len(h.layers))
        if self._deferred_gemms is not None:        # (short sequences: collected for the launch behind the last recurrence)
            head_grads()
        else:
            ops.stream_wait_value32(sync[0, 0][nchp - 1:nchp], da_target, stream=self.s_grad)
            with torch.cuda.stream(self.s_grad):
                head_grads()
        L = len(h.layers)
        if self._dec_kstream_ok(h.layers, B):
            # The stack's weight-gradient GEMMs as ONE K-streaming launch that FOLLOWS this launch chunk by chunk (as the encoder
            # stack's do, _encoder_backward_multi): they end with the BPTT instead of starting at its end, i.e. they no longer run
            # across the latent chain - a latency-bound kernel on the critical queue that takes 2-3 times as long beside them
            # (timeline r05_p: 156 us against 55-77 alone) - nor beside the encoder launch.
            cs, status = self.pipe_chunk, self.store["pipe_status"]
            problems = []
            for li, r in enumerate(reversed(h.layers)):
                problems += self._kstream_problems(r, B, None, dict(counters=sync[li, 0], target=da_target, rows=cs * B, status=status),
                                                   only_dU=r.xmode == hl.X_CONST)
            ops.stream_wait_value32(sync[L - 1, 0][nchp - 1:nchp], da_target, stream=self.s_grad2)
            with torch.cuda.stream(self.s_grad2):
                ops.gemm_kstream_multi(problems)
            return True
        for li, r in enumerate(reversed(h.layers)):
            if (li == L - 1 or self._hold_dec_grads >= 2) and self._hold_dec_grads and self._after_chain is not None:
                # the bottom layer's da is complete when the launch ends - exactly when the latent chain (a latency-bound kernel
                # between this phase and the next) starts: its gradient GEMMs wait for the ENCODER launch's first chunk instead
                self._after_chain.append(lambda r=r: dict(r=r, B=B, start=start))
            else:
                self._rec_param_grads(r, B, start=start, gate=(sync[li, 0][0:1], da_target))
        return True

    def _encoder_backward_multi(self, B, dcat, ldc, latent_grads=()):
        """BPTT of every encoder recurrence as one launch; the notes stack's weight gradients as the K-streaming launch behind it
        (engine._kstream_ok), the rest of the parameter gradients on the gradient queues behind one event"""
        s = self.spec
        H, T = s.H, s.T
        singles = [r for r, _, _ in self.enc_meta]
        if self.enc_bi or not self._phase_ok(self.enc_notes, singles):
            return False
        idx = self._v("in.x_idx", T, B)
        kstream = self._kstream_ok(self.enc_notes, B)
        probs, gemms, ks = self._stack_problems_backward(self.enc_notes, B, 3, dh_last=dcat[:, 0:H], dh_last_ld=ldc)
        sync, da_target, nchp = ks
        status = self.store["pipe_status"]
        extra, gates, follow, single_gate = [], [], None, {}
        for k, (r, _, src) in enumerate(self.enc_meta, 1):
            # every single-layer branch publishes its da too: its gradient work is released by ITS last chunk (a 4-step instrument
            # roll is done 1.8 ms before the launch ends), and a full-length one's dU can join the K-streaming launch
            cs = self.pipe_chunk if r.T % self.pipe_chunk == 0 else r.T
            sync1, target1, _ = self._sync_region(5 + k, 1, r.T // cs, self._rnn_waves(r) * (B // 16), 0)
            probs.append(self._rec_bptt(r, B, dh_last=dcat[:, k * H:(k + 1) * H], dh_last_ld=ldc, build=True,
                                        pipe=dict(chunk_steps=cs, status=status, signal_done=sync1[0, 0])))
            single_gate[r.prefix] = (sync1[0, 0][0:1], target1)
            if (kstream and follow is None and self.kstream_singles and r.T == T and cs == self.pipe_chunk and
                    (2 if s.cell == "GRU" else 1) + (3 if s.cell == "GRU" else 2) * len(self.enc_notes) <= 8):
                ks1 = dict(counters=sync1[0, 0], target=target1, rows=cs * B, status=status)
                extra = self._kstream_problems(r, B, None, ks1, only_dU=True)
                gates.append((sync1[0, 0][r.T // cs - 1:r.T // cs], target1))
                follow = r
        self._launch_phase_backward(("rnn_bwd_multi", "enc"), probs, gemms, steps=T * len(self.enc_notes))
        if latent_grads:        # the latent block's parameter gradients (ten small launches): the launch's first published chunk
            # says the chain is done; on a queue that is idle in this phase, not in front of the rolls' gradient work
            lq = self.s_instr if self.enc_instr is not None else self.s_grad
            ops.stream_wait_value32(sync[0, 0][nchp - 1:nchp], da_target, stream=lq)
            with torch.cuda.stream(lq):
                for fn in latent_grads:
                    fn()
            if lq is not self.s_grad:
                self._tail_streams.append(lq)
        held, self._after_chain = self._after_chain or [], None
        for fn in held:         # (decoder gradient work held back behind the latent chain: _notes_backward_multi)
            kw = fn()
            self._rec_param_grads(kw["r"], kw["B"], start=kw["start"], gate=(sync[0, 0][nchp - 1:nchp], da_target))
        if self._deferred_gemms is not None and self.defer_early and T * B >= self.defer_early_rows:
            # short sequences: the decoder side's collected weight-gradient GEMMs run BESIDE this launch instead of behind it
            L = len(self.enc_notes)
            self._flush_deferred_gemms(early=(sync[L - 1, 0][nchp - 1:nchp], da_target))
        if kstream:
            cs = self.pipe_chunk
            problems = []
            for li, r in enumerate(reversed(self.enc_notes)):
                problems += self._kstream_problems(r, B, idx, dict(counters=sync[li, 0], target=da_target, rows=cs * B, status=status))
            problems += extra
            for word, value in gates:
                ops.stream_wait_value32(word, value, stream=self.s_grad2)
            # (held back until the BOTTOM layer has published its first chunk: every kernel the workgroups wait for is resident then)
            L = len(self.enc_notes)
            ops.stream_wait_value32(sync[L - 1, 0][nchp - 1:nchp], da_target, stream=self.s_grad2)
            with torch.cuda.stream(self.s_grad2):
                ops.gemm_kstream_multi(problems)
            self._grad_streams = (self.s_grad, self.s_grad)       # the second gradient queue holds the K-streaming launch
        else:
            for li, r in enumerate(reversed(self.enc_notes)):
                self._rec_param_grads(r, B, idx=idx, gate=(sync[li, 0][0:1], da_target))
        for r, _, src in sorted(self.enc_meta, key=lambda m: m[0].T):        # (the short rolls first: they are done first)
            inp = (dict(xs=self._v(src, r.T, B)) if r.xmode == hl.X_SCALAR else dict(idx=self._v(src, r.T, B)))
            self._rec_param_grads(r, B, skip_dU=r is follow, gate=single_gate[r.prefix], **inp)
        self._grad_streams = None
        return True
