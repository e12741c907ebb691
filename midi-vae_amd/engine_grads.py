"""Engine mixin: the parameter-gradient work of a layer - which GEMMs and small reductions, on which queue, released by what
(reference: the autodiff of the layers of vae_definition.py:443-726 w.r.t. their kernels; Keras derives it, here it is scheduled by
hand: beside the recurrences for long sequences, as one batched launch behind the last recurrence for short ones)."""
from __future__ import annotations

import contextlib

import torch

from . import hiplib as hl
from . import ops
from .slots import X_EXT, X_GATHER2



def kstream_parts(tiles, chunk_rows, kstream_rows, kstream_wgs):
    """K partitions (a power of two) of one chunk of a K-streaming weight-gradient GEMM with ``tiles`` 128 x 128 output tiles; its launch
    gets tiles * partitions workgroups, each a whole number of 64-row k tiles per chunk.
    ``kstream_rows`` > 0 (round 6): every workgroup of the launch gets about that many k rows per chunk, whatever its GEMM's tile count
    - never more than 1.5 x ``kstream_wgs`` workgroups for one GEMM (the launch's workgroups all wait resident).  With ``kstream_wgs``
    workgroups per GEMM (the round-2 rule, ``kstream_rows`` = 0) the 12-tile dW of a dense GRU layer ran one tile and a whole chunk per
    workgroup (4 of 16 idle) beside 4-tile GEMMs split four ways: that problem paced the launch, which ended 0.23 ms behind the encoder
    BPTT once the two-waves-per-SIMD kernels had made the BPTT faster."""
    P = 1
    if kstream_rows:
        while (chunk_rows // (P * 2) >= kstream_rows and chunk_rows % (P * 2 * 64) == 0 and tiles * P * 2 <= kstream_wgs * 3 // 2):
            P *= 2
        return P
    while P * 2 * tiles <= kstream_wgs and chunk_rows % (P * 2 * 64) == 0:
        P *= 2
    return P

class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ParamGradients(object):
    def _grad_portions(self, r, B, publishes):
        """Time portions of a layer's parameter-gradient work (per-queue schedule, LONG sequences).  At T * B >= 2^19 rows the
        gradient GEMMs of a phase are milliseconds of whole-chip work that used to start when the layer's BPTT ENDS - at BASELINE
        configs[2]'s shape (T=2048, 512 windows) 4 ms of tail behind the last BPTT and a latent-chain kernel starved for 1.5 ms
        between the two BPTT phases (profiles/r04_m_timeline_config2_lstm.txt).  A layer that publishes its da chunks
        (``publishes`` = (counter array, target, chunk steps)) has them released in P portions of the time axis by the chunk that
        completes each portion (hipStreamWaitValue32 on the gradient queues): each portion is a full-size GEMM here (>= 2^17 rows:
        round 2 tried it at T=512 x 256 windows, where a portion was all atomic epilogue), and only the last one is left when the
        recurrence ends.  Portions of all layers of a phase are enqueued portion-major (_flush_grad_portions): a queue parked on
        one layer's last chunk must not hold another layer's first portions."""
        if publishes is None or self._grad_portion_jobs is None:
            return 1
        return self._portion_count(r, B, publishes[2])

    def _portion_count(self, r, B, cs):
        if not self.grad_portions:
            return 1
        R = r.T * B
        P = self.grad_portions if self.grad_portions > 0 else int(min(4, R // (1 << 18)))
        while P > 1 and (r.T % P or (r.T // P) % cs):
            P -= 1
        return max(P, 1)

    def _flush_grad_portions(self):
        """launch the collected gradient portions of a phase, portion-major (first portions of every layer first)"""
        jobs, self._grad_portion_jobs = self._grad_portion_jobs, None
        for _, fn in sorted(jobs or [], key=lambda j: j[0]):
            fn()

    def _wgemm(self, A, Bm, C, M, N, K, **kw):
        """a weight-gradient GEMM C (M,N) f32 += A^T Bm (ops.gemm with trans_a, accumulate): launched now on the current stream -
        or, on a step that defers them (defer_grads_rows), kept as a problem of the one mvae_gemm_multi launch behind the last
        recurrence"""
        if self._deferred_gemms is not None and self.tile16 and K % 64 == 0 and (N % 128 == 0 or N < 128):
            self._deferred_gemms.append(ops.gemm(A, Bm, C, M, N, K, trans_a=True, accumulate=True, build_only=True, **kw))
            return
        ops.gemm(A, Bm, C, M, N, K, trans_a=True, accumulate=True, **kw)

    def _small(self, fn):
        """a small launch of the parameter-gradient work (a column sum, a sum over time): now on the current stream - or, on a
        step that defers (defer_grads_rows), behind the batched GEMM launch on the critical queue (a gate + a launch on a gradient
        queue costs the command processors more than these kernels run)"""
        if self._deferred_gemms is not None:
            self._deferred_small.append(fn)
        else:
            fn()

    def _flush_deferred_gemms(self, early=None):
        """the step's collected weight-gradient GEMMs as one launch (per 16) on the current - the critical - stream, the small
        launches behind it: everything they read was produced on this queue or joined into it.
        ``early`` = (counter word, value): what has been collected SO FAR (the decoder side, the latent block) leaves now, on the
        gradient queue, released on the device by the running encoder launch's first published chunk - its workgroups are
        resident then, the GEMMs take the CUs it left empty - and the collection goes on for the encoder's own gradients."""
        probs, small = self._deferred_gemms, self._deferred_small
        self._deferred_gemms, self._deferred_small = ([] if early is not None else None), []
        if early is not None:
            if not probs and not small:
                return
            ops.stream_wait_value32(early[0], early[1], stream=self.s_grad)
        if probs and self.defer_split_wgs:
            # a short sequence's GEMMs are few workgroups with long k loops (T*B = 16384 rows: split_k 2, ~150 workgroups of 128 k
            # tiles each for the whole launch): split K further until the launch has about defer_split_wgs workgroups
            tiles = [(-(-g.M // 128)) * (-(-g.N // 128)) for g in probs]
            total = sum(t * max(1, g.split_k) for t, g in zip(tiles, probs))
            f = 1
            while total * f * 2 <= self.defer_split_wgs and f < 8:
                f *= 2
            for g in probs:
                sk = max(1, g.split_k) * f
                while sk > 1 and g.K // sk < 512:
                    sk //= 2
                g.split_k = min(16, sk)
        with (torch.cuda.stream(self.s_grad) if early is not None else contextlib.nullcontext()):
            if probs:
                # (a shape the batched launch does not take: that part and what follows one by one - NOT the parts in front of it,
                #  which have been launched and accumulate)
                for g in probs[ops.gemm_multi(probs):]:
                    ops.gemm_args(g)
            for fn in small:
                fn()

    def _rec_param_grads(self, r, B, *, idx=None, xs=None, start=None, skip_dU=False, gate=None, publishes=None, rows=None):
        P = self._grad_portions(r, B, publishes) if rows is None else 1
        if P > 1:
            counters, target, cs = publishes
            Tp = r.T // P
            for pi in range(P):                      # BPTT order: the LAST steps first
                t_lo = r.T - (pi + 1) * Tp
                g = (counters[t_lo // cs:t_lo // cs + 1], target)
                self._grad_portion_jobs.append((pi, lambda t_lo=t_lo, g=g, pi=pi: self._rec_param_grads_rows(
                    r, B, idx=idx, xs=xs, start=start, skip_dU=skip_dU, gate=g, rows=(t_lo, t_lo + Tp), first=pi == 0)))
            return
        return self._rec_param_grads_rows(r, B, idx=idx, xs=xs, start=start, skip_dU=skip_dU, gate=gate, rows=rows)

    def _rec_param_grads_rows(self, r, B, *, idx=None, xs=None, start=None, skip_dU=False, gate=None, rows=None, first=True):
        """Parameter gradients of one layer from its da, accumulated into the f32 gradient buffer: off the critical path, on
        the two gradient streams, once per layer after its BPTT.  ``gate`` = (counter word, value): the layer's BPTT is a problem
        of a phase launch that is still RUNNING - the gradient queues wait, on the device, for the layer's last published chunk
        of da instead of for the whole launch.  (No event: the launch sits on the critical queue behind everything the gradient
        work reads, so its first published chunk implies all of that; an event record would be one more packet there.)"""
        if self._diag_no_param_grads:       # (MVAE_DIAG_NO_PARAM_GRADS=1, timing experiments only: the gradients are WRONG)
            return
        s, G, p = self.spec, self.G, r.prefix
        H, GH, T = s.H, s.GH, r.T
        t_lo, t_hi = rows if rows is not None else (0, T)        # (a time portion: the same GEMMs over rows [t_lo * B, t_hi * B))
        Tq = t_hi - t_lo
        R = Tq * B
        da = self._v(p + ".da", T, B, GH)[t_lo:t_hi]
        da2, hprev = da.view(R, GH), self._v(p + ".hs", T + 1, B, H)[t_lo:t_hi].reshape(R, H)
        if idx is not None:
            idx = idx[t_lo:t_hi]
        if xs is not None:
            xs = xs[t_lo:t_hi]
        sk = self._split_k(R)
        sg1, sg2 = self._grad_streams or (self.s_grad, self.s_grad2)
        deferring = self._deferred_gemms is not None
        if deferring:                       # (everything below is collected: nothing is enqueued on the gradient queues now)
            on1 = on2 = _NullCtx()
        else:
            on1, on2 = self._on(sg1), self._on(sg2)
            if gate is None:
                self._fork(*((sg1,) if sg1 is sg2 else (sg1, sg2)))
            else:
                for st in ((sg1,) if sg1 is sg2 else (sg1, sg2)):
                    ops.stream_wait_value32(gate[0], gate[1], stream=st)
        # bias gradient = column sums of da: from the recurrent-kernel gradient GEMM's own pass over da (fast bf16 path) - also for
        # the decoder cells on a constant input when that input is all zeros (what the reference's packers always pass,
        # vae_definition.py:820,916: dW = start^T sum_t(da) = 0 then, and the sum over time is only needed for the bias)
        const_fused = (r.xmode == hl.X_CONST and self.start_zero.get(p, False) and self.tile16 and self.fuse_bias_grad and
                       not skip_dU)
        fuse_b = (r.xmode != hl.X_CONST or const_fused) and self.tile16 and self.fuse_bias_grad
        gb = G[p + ".b"]
        with on1:
            # recurrent kernel: dU = sum_t h_{t-1}^T da_t   (GRU candidate block uses r*h_{t-1})
            if skip_dU:             # (with its bias gradient in a K-streaming launch)
                pass
            elif s.cell == "GRU":
                rh = self._v(p + ".rh", T, B, H)[t_lo:t_hi]
                self._wgemm(hprev, da2, G[p + ".U"], H, 2 * H, R, ldb=GH, ldc=GH, split_k=sk, colsum_b=gb[:2 * H] if fuse_b else None)
                self._wgemm(rh.view(R, H), da2[:, 2 * H:], G[p + ".U"][:, 2 * H:], H, H, R, ldb=GH, ldc=GH, split_k=sk,
                            colsum_b=gb[2 * H:] if fuse_b else None)
            else:
                self._wgemm(hprev, da2, G[p + ".U"], H, GH, R, split_k=sk, colsum_b=gb if fuse_b else None)
        with on2:
            if const_fused:
                pass
            elif r.xmode == hl.X_CONST:
                dxp0 = self._v(p + ".dxp0", B, GH)
                acc = self._dxp0_clean or not first
                # (time portions: every portion adds its share to dxp0 - zeroed by the weight preparation; what is derived from the
                #  complete sum follows the portion that ends at step 0)
                self._small(lambda: ops.sum_over_time(da, Tq, B * GH, dxp0, accumulate=acc))
                if t_lo == 0:
                    self._small(lambda: ops.colsum(dxp0, B, GH, G[p + ".b"]))
                    if not self.start_zero.get(p, False):        # (dW = start^T dxp0 = 0 for an all-zero start)
                        self._small(lambda: ops.gemm(start, dxp0, G[p + ".W"], r.K, GH, B, trans_a=True, accumulate=True))
            else:
                if not fuse_b:
                    self._small(lambda: ops.colsum(da2, R, GH, G[p + ".b"]))
                if r.xmode == X_EXT:
                    pass                                # (input-kernel gradient by the caller: _aux_backward)
                elif r.xmode == hl.X_INDEX:
                    self._wgemm(idx.reshape(-1), da2, G[p + ".W"], r.K, GH, R, a_kind=hl.ONEHOT, split_k=sk)
                elif r.xmode == X_GATHER2:      # two-hot rows: the pitch rows and the attached instrument rows of W
                    d0 = r.K - s.attach
                    self._wgemm(idx.reshape(-1), da2, G[p + ".W"][:d0], d0, GH, R, a_kind=hl.ONEHOT, split_k=sk)
                    self._wgemm(self._v("in.xa_idx", T, B)[t_lo:t_hi].reshape(-1), da2, G[p + ".W"][d0:], s.attach, GH, R,
                                a_kind=hl.ONEHOT, split_k=sk)
                elif r.xmode == hl.X_SCALAR:       # dW (1, GH) = xs^T da: a weighted column sum
                    self._small(lambda: ops.colsum_weighted(da2, xs.reshape(-1), R, GH, G[p + ".W"]))
                else:
                    lower = self._v(r.lower.prefix + ".hs", T + 1, B, H)[1 + t_lo:1 + t_hi].reshape(R, H)
                    self._wgemm(lower, da2, G[p + ".W"], H, GH, R, split_k=sk)

    def _kstream_ok(self, layers, B):
        """K-streaming weight-gradient GEMMs behind the BPTT kernels of this stack?  Every layer's gradients must be GEMMs (index
        or dense input, bias gradient fused), at most 8 of them - and their workgroups, which wait RESIDENT for the whole BPTT,
        must find their CUs beside the stack's own kernels and the phase's other recurrences: a workgroup that only gets its CU
        when another one retires does its whole share after the recurrence, which is the tail the launch exists to remove.
        (At 256 windows: 256 - (32 + 32 + 32) = 160 free CUs for 128 workgroups - the single-layer branch's 32 on top are the ones
        that may start late; at 512 windows 96: ordinary GEMMs then, as measured, profiles/r02_q_pipe_chunk_by_batch.txt.)"""
        s = self.spec
        count = (3 if s.cell == "GRU" else 2) * len(layers)      # GEMMs per layer: dU (GRU: two launches) and dW
        if not (self.kstream_grads and self._deferred_gemms is None and self.multi_stream and self.fuse_bias_grad and self.tile16 and count <= 8 and
                self._pipelined(layers) and all(r.xmode in (hl.X_INDEX, hl.X_DENSE) for r in layers)):
            return False
        if B > self.kstream_max_B:       # (a chunk's rows grow with the batch, the time the BPTT takes for it does not)
            return False
        free = (self.num_cus - self._resident_cus(layers, B, backward=True, side=True)) * self._occ["kstream"]
        return free >= self.kstream_wgs * count

    def _dec_kstream_ok(self, layers, B):
        """the decoder notes stack's weight gradients as a K-streaming launch behind its BPTT launch?  As _kstream_ok, for a stack
        whose bottom cell steps on a constant ALL-ZERO start row (what the reference's packers pass, vae_definition.py:820,916: its
        input-kernel gradient is zero and its bias gradient the column sums of da, fused into the dU GEMM)."""
        s = self.spec
        if not (self.dec_kstream and self.kstream_grads and self._deferred_gemms is None and self.multi_stream and
                self.fuse_bias_grad and self.tile16 and self._pipelined(layers) and B <= self.kstream_max_B and not self._diag_no_param_grads):
            return False
        bottom, rest = layers[0], layers[1:]
        if not (bottom.xmode == hl.X_CONST and self.start_zero.get(bottom.prefix, False) and all(r.xmode == hl.X_DENSE for r in rest)):
            return False
        per = 2 if s.cell == "GRU" else 1
        count = per * len(layers) + len(rest)
        if count > 8:
            return False
        free = (self.num_cus - self._resident_cus(layers, B, backward=True, side=True)) * self._occ["kstream"]
        return free >= self.kstream_wgs * count

    def _kstream_problems(self, r, B, idx, ks, only_dU=False):
        """the layer's weight-gradient GEMMs as K-streaming problems (mvae_gemm_args, not launched)"""
        s, G, p = self.spec, self.G, r.prefix
        H, GH, T = s.H, s.GH, r.T
        R = T * B
        hprev = self._v(p + ".hs", T + 1, B, H)[:T].reshape(R, H)
        da2 = self._v(p + ".da", T, B, GH).view(R, GH)
        kw = dict(k_wait=ks["counters"], k_wait_value=ks["target"], k_chunk_rows=ks["rows"], k_reverse=True, chunk_status=ks["status"],
                  trans_a=True, accumulate=True, build_only=True)
        def parts(M, N):        # K partitions per chunk, whole 64-row k tiles each
            return kstream_parts(-(-M // 128) * -(-N // 128), ks["rows"], self.kstream_rows, self.kstream_wgs)
        gb = G[p + ".b"]
        out = []
        if s.cell == "GRU":
            rh = self._v(p + ".rh", T, B, H)
            out.append(ops.gemm(hprev, da2, G[p + ".U"], H, 2 * H, R, ldb=GH, ldc=GH, split_k=parts(H, 2 * H),
                                colsum_b=gb[:2 * H], **kw))
            out.append(ops.gemm(rh.view(R, H), da2[:, 2 * H:], G[p + ".U"][:, 2 * H:], H, H, R, ldb=GH, ldc=GH, split_k=parts(H, H),
                                colsum_b=gb[2 * H:], **kw))
        else:
            out.append(ops.gemm(hprev, da2, G[p + ".U"], H, GH, R, split_k=parts(H, GH), colsum_b=gb, **kw))
        if only_dU:
            return out
        if r.xmode == hl.X_INDEX:
            out.append(ops.gemm(idx.reshape(-1), da2, G[p + ".W"], r.K, GH, R, a_kind=hl.ONEHOT, split_k=parts(r.K, GH), **kw))
        else:
            lower = self._v(r.lower.prefix + ".hs", T + 1, B, H)[1:1 + T].reshape(R, H)
            out.append(ops.gemm(lower, da2, G[p + ".W"], H, GH, R, split_k=parts(H, GH), **kw))
        return out

