"""Host-array staging and result read-back of the device engine (mixins of engine.Engine).

``ArrayStaging``: the engine's inputs from host NumPy arrays in the ENGINE's own format (one byte per row, f32 rolls) - what
bench.py, the tools and the kernel-level tests use; the reference-format float64 lists go through staging.Stager instead.
``Results``: losses / metrics (per step and accumulated per epoch), decoder outputs, the pipeline status check, profiling
summaries.  Reference sites on the methods.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import hiplib as hl
from . import ops
from .slots import *        # noqa: F401,F403
from .slots import N_SCALARS


class ArrayStaging(object):
    def _note_start(self, name, val):
        """remember whether the start rows staged into ``name`` (in.start_<head>) are all zero: the head's first cell then reads the
        bias rows the weight preparation wrote (no start*W GEMM, no dW GEMM)"""
        h = self.head.get(name[len("in.start_"):])
        if h is not None:
            self.start_zero[h.layers[0].prefix] = val is None or not np.any(val)

    # ------------------------------------------------------------------------------------------------------
    # input staging (host NumPy -> device).  Layout conversion to time-major happens here, once, on the host.
    # ------------------------------------------------------------------------------------------------------
    @staticmethod
    def pad16(B):
        return (int(B) + 15) // 16 * 16

    def _up(self, name, arr, tdtype):
        a = np.ascontiguousarray(arr)
        t = torch.from_numpy(a).to(self.device, non_blocking=False).to(tdtype)
        self.store[name][:t.numel()].copy_(t.reshape(-1))

    def _up_tm(self, name, arr_bt, tdtype, fill=0):
        """(B, L) batch-major host array -> (L, Bp) time-major device buffer, pad rows = ``fill``."""
        arr_bt = np.asarray(arr_bt)
        B, L = arr_bt.shape
        out = np.full((L, self.pad16(B)), fill, dtype=arr_bt.dtype)
        out[:, :B] = arr_bt.T
        self._up(name, out, tdtype)

    def _up_rows(self, name, arr, width, tdtype=torch.float32):
        """(B, width) host array -> first B rows of the (Bp, width) device buffer; pad rows zeroed."""
        arr = np.asarray(arr).reshape(-1, width)
        B = arr.shape[0]
        out = np.zeros((self.pad16(B), width), dtype=arr.dtype)
        out[:B] = arr
        self._up(name, out, tdtype)

    def stage_encoder_inputs(self, x_idx, i_idx=None, vel=None, eps=None, d_idx=None, xa_idx=None):
        """x_idx (B,T) uint8 note index per row; i_idx (B,V) uint8; vel (B,T) f32; d_idx (B,T) uint8 held-notes flag; eps (B,Z)
        f32 ALREADY scaled by epsilon_std (None -> zeros: deterministic encode, like the evaluation script's epsilon_std = 0)."""
        B = x_idx.shape[0]
        self.norm_B = float(B)
        self._up_tm("in.x_idx", np.asarray(x_idx, np.uint8), torch.uint8)
        if self.spec.attach:            # (B,T) uint8: the attached instrument column of every row, 0 .. attach-1
            self._up_tm("in.xa_idx", np.asarray(xa_idx, np.uint8), torch.uint8)
        if self.enc_bi:
            self._up_tm("in.x_idx_rev", np.asarray(x_idx, np.uint8)[:, ::-1], torch.uint8)
        if self.spec.meta_instrument:
            self._up_tm("in.i_idx", np.asarray(i_idx, np.uint8), torch.uint8)
        if self.spec.meta_velocity:
            self._up_tm("in.vel", np.asarray(vel, np.float32), torch.float32)
        if self.spec.meta_held:
            self._up_tm("in.d_idx", np.asarray(d_idx, np.uint8), torch.uint8)
        self._up_rows("in.eps", np.zeros((B, self.spec.Z), np.float32) if eps is None else np.asarray(eps, np.float32),
                      self.spec.Z)
        return B

    def stage_decoder_inputs(self, B, hist=None, z=None, start_notes=None, start_instr=None, start_vel=None, start_held=None,
                             start_next=None, add=None):
        s = self.spec
        Bp = self.pad16(B)
        zh = self._v("zh", Bp, s.zin)
        zh[B:].zero_()
        if s.history:
            if hist is None:
                zh[:, s.Z:2 * s.Z].zero_()
            else:
                zh[:B, s.Z:2 * s.Z].copy_(torch.from_numpy(np.ascontiguousarray(hist, np.float32)).to(self.device))
        if z is not None:
            zh[:B, :s.Z].copy_(torch.from_numpy(np.ascontiguousarray(z, np.float32)).to(self.device))
        if s.add_dim:           # the decoder's additional input (reference vae_definition.py:553-556): behind [z | history]
            a0 = s.zin - s.add_dim
            if add is None:
                zh[:, a0:].zero_()
            else:
                zh[:B, a0:].copy_(torch.from_numpy(np.ascontiguousarray(add, np.float32).reshape(B, s.add_dim)).to(self.device))
        for name, val, width in (("in.start_notes", start_notes, s.Dout), ("in.start_instr", start_instr, s.ID),
                                 ("in.start_vel", start_vel, 1), ("in.start_held", start_held, 2),
                                 ("in.start_next", start_next, s.Dout)):
            if name not in self.store:
                continue
            self._up_rows(name, np.zeros((B, width), np.float32) if val is None else np.asarray(val, np.float32), width)
            self._note_start(name, val)

    def stage_targets(self, B, y_idx, c_idx=None, w_notes=None, w_instr=None, w_vel=None, w_style=None, n_idx=None,
                      w_held=None, w_next=None, sig=None, w_sig=None, w_cnotes=None, w_cinstr=None, ya_idx=None):
        """Targets and Keras sample weights.  Row weights are folded with the weighted-objective normalisers
        (score*w / mean(w != 0), then the mean over the axes) into one factor per row; padding
        rows get target 255 ("no target") and weight 0."""
        s = self.spec
        T, V = s.T, s.V
        self.norm_B = float(B)
        self._up_tm("in.y_idx", np.asarray(y_idx, np.uint8), torch.uint8, fill=255)
        if s.attach:                    # second hot column of the two-hot target rows, as an ABSOLUTE column (D0 + instrument)
            self._up_tm("in.ya_idx", (np.asarray(ya_idx, np.int64) + (s.Dout - s.attach)).astype(np.uint8), torch.uint8, fill=255)

        def norm(w, n_other):
            w = np.asarray(w, np.float64)
            nz = np.mean(w != 0)
            return (w / (nz * w.size * n_other)).astype(np.float32)

        wn = np.ones((B, T)) if w_notes is None else w_notes
        self._up_tm("in.rw_notes", norm(wn, 1), torch.float32)
        if s.meta_instrument:
            wi = np.ones((B,)) if w_instr is None else w_instr
            self._up_tm("in.rw_instr", np.repeat(norm(wi, V)[:, None], V, axis=1), torch.float32)
        if s.meta_velocity:
            wv = np.ones((B,)) if w_vel is None else w_vel
            self._up_tm("in.rw_vel", np.repeat(norm(wv, T)[:, None], T, axis=1), torch.float32)
        if s.meta_held:          # (the target is the held-notes roll staged with the encoder inputs)
            wh = np.ones((B,)) if w_held is None else w_held
            self._up_tm("in.rw_held", np.repeat(norm(wh, T)[:, None], T, axis=1), torch.float32)
        if s.meta_next:
            wx = np.ones((B,)) if w_next is None else w_next
            self._up_tm("in.rw_next", np.repeat(norm(wx, T)[:, None], T, axis=1), torch.float32)
            self._up_tm("in.n_idx", np.asarray(n_idx, np.uint8), torch.uint8, fill=255)
        if s.style:
            ws = np.ones((B,)) if w_style is None else w_style
            self._up("in.rw_style", norm(ws, 1), torch.float32)
        if s.style or self.aux:
            c = np.full((self.pad16(B),), 255, np.uint8)       # (padding rows: "no target")
            c[:B] = np.asarray(c_idx, np.uint8)
            self._up("in.c_idx", c, torch.uint8)
        if s.signature:
            self._up_rows("in.sig", np.asarray(sig, np.float32), s.SD)
            self._up_rows("in.rw_sig", norm(np.ones((B,)) if w_sig is None else w_sig, 1), 1)
        for a, w in zip(self.aux, [w_cnotes if a.key == "cnotes" else w_cinstr for a in self.aux]):
            self._up_rows("in.rw_" + a.key, norm(np.ones((B,)) if w is None else w, 1), 1)


class Results(object):
    # hit-count slots of the scalar block (accumulated as counts; everything else as batch-size weighted means)
    HIT_MASK = ((1 << S_NOTES_HITS) | (1 << S_INSTR_HITS) | (1 << S_VEL_HITS) | (1 << S_STYLE_HITS) | (1 << S_HELD_HITS) |
                (1 << S_NEXT_HITS) | (1 << S_SIG_HITS) | (1 << S_CNOTES_HITS) | (1 << S_CINSTR_HITS))

    def reset_accumulated(self):
        """start a fresh set of epoch accumulators (a NEW device buffer: a History that has not been read yet keeps its own)"""
        self.acc = torch.zeros(N_SCALARS, dtype=torch.float32, device=self.device)
        return self.acc

    def accumulate_metrics(self, B_global):
        """acc += B_global * (loss slots), += (hit slots) of the step just enqueued - Keras' BaseLogger on the device, no read"""
        ops.scalars_accumulate(self.acc, self.scal, float(B_global), self.HIT_MASK)

    def read_accumulated(self, n_windows, allreduce_sum=None, acc=None):
        """means over ``n_windows`` windows of everything accumulated into ``acc`` (default: since the last reset_accumulated): ONE
        device->host read; with ``allreduce_sum`` the per-rank shares are summed first"""
        acc = self.acc if acc is None else acc
        if allreduce_sum is not None:
            allreduce_sum(acc)
        v = acc.cpu().numpy().astype(np.float64)
        self.check_pipeline()
        n = max(float(n_windows), 1.0)
        hit = np.array([(self.HIT_MASK >> i) & 1 for i in range(N_SCALARS)], bool)
        v = np.where(hit, v, v / n)
        return self._metrics_from(v, n)

    # ------------------------------------------------------------------------------------------------------
    # results
    # ------------------------------------------------------------------------------------------------------
    def check_pipeline(self):
        """Raises if a kernel of a time-pipelined stack gave up waiting for its input since the last check (the results of that
        step are invalid; its optimizer update was skipped)."""
        code = int(self.store["pipe_words"].max().item())
        if code != 0:
            self.store["pipe_words"].zero_()
            kind = {1: "recurrent forward kernel", 2: "BPTT kernel", 3: "chunked GEMM", 4: "K-streaming GEMM", 5: "join"}.get(code, "kernel")
            raise RuntimeError("a device-side wait timed out (%s waiting for its producer: stream / hardware queue aliasing?); "
                               "set Engine.pipeline = False" % kind)

    def metrics(self, B) -> "OrderedDict[str, float]":
        """Losses / accuracies of the last step with the oracle's key names (one device->host copy)."""
        s = self.spec
        v = self.scal.cpu().numpy().astype(np.float64)
        self.check_pipeline()
        if B is not None and self.norm_B != B:
            B = self.norm_B             # a shard of a global minibatch: this rank's SHARE of the global means
        return self._metrics_from(v, B)

    def _metrics_from(self, v, B):
        """metric dict from the scalar slots: loss slots hold batch means already, hit slots counts over ``B`` windows"""
        s = self.spec
        m = OrderedDict()
        m["kl"] = v[S_KL]
        m["notes_loss"], m["notes_acc"] = v[S_NOTES_LOSS], v[S_NOTES_HITS] / (B * s.T)
        total = m["notes_loss"] + m["kl"]
        if s.meta_instrument:
            m["instr_loss"], m["instr_acc"] = v[S_INSTR_LOSS], v[S_INSTR_HITS] / (B * s.V)
            total += s.w_instr * m["instr_loss"]
        if s.meta_velocity:
            m["vel_loss"], m["vel_acc"] = v[S_VEL_LOSS], v[S_VEL_HITS] / (B * s.T)
            total += s.w_vel * m["vel_loss"]
        if s.meta_held:
            m["held_loss"], m["held_acc"] = v[S_HELD_LOSS], v[S_HELD_HITS] / (B * s.T)
            total += s.w_held * m["held_loss"]
        if s.meta_next:
            m["next_loss"], m["next_acc"] = v[S_NEXT_LOSS], v[S_NEXT_HITS] / (B * s.T)
            total += s.w_next * m["next_loss"]
        if s.style:
            m["style_loss"], m["style_acc"] = v[S_STYLE_LOSS], v[S_STYLE_HITS] / B
            total += s.w_style * m["style_loss"]
        if s.signature:
            m["sig_loss"], m["sig_acc"] = v[S_SIG_LOSS], v[S_SIG_HITS] / B
            total += s.w_sig * m["sig_loss"]
        for a in self.aux:
            m[a.key + "_loss"], m[a.key + "_acc"] = v[a.slot], v[a.slot + 1] / B
            total += a.weight * m[a.key + "_loss"]
        m["loss"] = total
        return m

    def outputs(self, B):
        """Batch-major NumPy copies of the decoder outputs of the last forward run with want_probs=True."""
        s = self.spec
        Bp = self.pad16(B)
        out = OrderedDict()
        for h in self.dec_heads:
            out[h.name] = self._v("out.%s_p" % h.name, h.T, Bp, h.N)[:, :B].permute(1, 0, 2).cpu().numpy()
        if s.style:
            out["style"] = self._v("style_p", Bp, s.C)[:B].cpu().numpy()
        if s.signature:
            out["sig"] = self._v("sig.out", Bp, s.SD)[:B].cpu().numpy()
        for a in self.aux:
            out[a.key] = self._v("out.%s_p" % a.key, Bp, s.C)[:B].cpu().numpy()
        return out

    def note_indices(self, B):
        """(B,T) uint8 argmax note index per row - the fused form of sample_vector(...,'argmax')."""
        return self._v("notes.argmax", self.spec.T, self.pad16(B))[:, :B].t().contiguous().cpu().numpy()

    def latent(self, B):
        return self._v("zh", self.pad16(B), self.spec.zin)[:B, :self.spec.Z].cpu().numpy()

    def prof_summary(self):
        """key -> (launches, mean ms, mean time steps per launch) for the event pairs collected since
        ``self.prof = {}`` (synchronises first)."""
        torch.cuda.synchronize()
        for i in list(self._prof_pending):
            self._prof_harvest(i)
        return {k: (len(v), float(np.mean([ms for ms, _ in v])), float(np.mean([n for _, n in v]))) for k, v in (self.prof or {}).items()}

    def bytes_resident(self):
        return (sum(t.numel() * t.element_size() for t in self.store.values()) +
                4 * self.params.numel() * 4)

    @staticmethod
    def forward_bytes_per_window(spec, kind):
        """HBM a forward-only engine holds per window of its batch (sizing of model._Shared.get_infer): the h sequences of every
        recurrent layer, x*W + b of the stacked / 1-feature layers, the heads' probabilities on request, inputs"""
        e = 2 if kind == hl.BF16 else 4
        T, V, H, GH = spec.T, spec.V, spec.H, spec.GH
        n_T = spec.Le + spec.Ld + 2 * int(spec.meta_velocity) + 2 * int(spec.meta_held) + spec.Ld * int(spec.meta_next)
        n_xp = (spec.Le - 1) + (spec.Ld - 1) + int(spec.meta_velocity) + (spec.Ld - 1) * int(spec.meta_next)
        per = n_T * (T + 1) * H * e + 2 * int(spec.meta_instrument) * (V + 1) * H * e + n_xp * T * GH * e
        per += T * (spec.Dout * 4 + 64) + 4096
        return int(per * 1.25)
