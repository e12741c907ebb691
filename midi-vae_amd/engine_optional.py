"""Parts of the graph that are off in the reference's default configuration, and the one-launch-per-operation forms of the latent
block (mixin of engine.Engine): the bidirectional encoder stack (reference vae_definition.py:445-453), the style classifiers on the
decoder's outputs (:747-761), the signature head (:737-745), and the Dense chain around the latent as separate launches (parity
mode / shapes the fused chain of csrc/latent.hip does not take)."""
from __future__ import annotations

from . import hiplib as hl
from . import ops
from .slots import *        # noqa: F401,F403


class OptionalGraph(object):
    # ---- bidirectional encoder stack (reference vae_definition.py:445-453) --------------------------------------------------
    def _enc_bi_forward(self, B, h_last, ldc):
        """Le-2 Bidirectional(concat) layers and one plain layer on top.  The backward RNN of a pair runs the same kernels on
        the time-reversed input (reversed index roll for the one-hot layer, ``cat_rev`` above it); layer l+1 reads the
        time-aligned concatenation [forward | backward] of layer l (mvae_bi_concat), projected by ONE GEMM with K = 2H."""
        s, P = self.spec, self.P
        H, GH, T = s.H, s.GH, s.T
        R = T * B
        for li, layer in enumerate(self.enc_bi):
            top = li == len(self.enc_bi) - 1
            if li > 0:
                lo = self.enc_bi[li - 1]
                catb = self._v("enc.bi.%d.cat" % li, R, 2 * H)
                rev = self._v("enc.bi.%d.cat_rev" % li, R, 2 * H) if len(layer) > 1 else None
                ops.bi_concat(self._v(lo[0].prefix + ".hs", T + 1, B, H)[1:], self._v(lo[1].prefix + ".hs", T + 1, B, H)[1:], catb, rev,
                              T, B, H)
            for j, r in enumerate(layer):
                if li == 0:
                    self._rec_forward(r, B, idx=self._v("in.x_idx_rev" if j else "in.x_idx", T, B))
                else:
                    src = rev if j else catb
                    ops.gemm(src, self._v(r.prefix + ".wt2", GH, 2 * H), self._v(r.prefix + ".xp", R, GH), R, GH, 2 * H, trans_b=True,
                             bias=P[r.prefix + ".b"], c_layout=self.lay)
                    self._rec_forward(r, B, h_last=h_last if top else None, h_last_ld=ldc if top else 0)

    def _enc_bi_backward(self, B, dh_last, ldc):
        s, P, G = self.spec, self.P, self.G
        H, GH, T = s.H, s.GH, s.T
        R = T * B
        for li in range(len(self.enc_bi) - 1, -1, -1):
            layer = self.enc_bi[li]
            top = li == len(self.enc_bi) - 1
            for j, r in enumerate(layer):
                dext = None
                if not top:
                    dext = self._v("enc.bi.%d.dext_%s" % (li + 1, "r" if j else "f"), T, B, H)
                idx = self._v("in.x_idx_rev" if j else "in.x_idx", T, B) if li == 0 else None
                self._rec_bptt(r, B, dhs_ext=dext, dh_last=dh_last if top else None, dh_last_ld=ldc if top else 0)
                self._rec_param_grads(r, B, idx=idx)
                if li > 0:
                    da = self._v(r.prefix + ".da", R, GH)
                    src = self._v("enc.bi.%d.cat%s" % (li, "_rev" if j else ""), R, 2 * H)
                    self._side(lambda src=src, da=da, r=r: ops.gemm(src, da, G[r.prefix + ".W"], 2 * H, GH, R, trans_a=True,
                                                                    accumulate=True, split_k=self._split_k(R)))
                    wc = self._v(r.prefix + ".wc2", 2 * H, GH)
                    for half, name in ((0, ".g1"), (1, ".g2")):       # d(input)[:, :H] -> forward layer below, [:, H:] -> backward
                        ops.gemm(da, wc[half * H:(half + 1) * H], self._v(r.prefix + name, R, H), R, H, GH, trans_b=True,
                                 c_layout=self.lay)
            if li > 0:
                # the layer below: its forward RNN lives in natural time, its backward RNN in reversed time; gradients computed by
                # this layer's forward record are in natural time, by its backward record in reversed time
                f = layer[0].prefix
                g1f, g2f = self._v(f + ".g1", R, H), self._v(f + ".g2", R, H)
                df, dr = self._v("enc.bi.%d.dext_f" % li, R, H), self._v("enc.bi.%d.dext_r" % li, R, H)
                if len(layer) > 1:
                    b = layer[1].prefix
                    g1b, g2b = self._v(b + ".g1", R, H), self._v(b + ".g2", R, H)
                    ops.add_time_reversed(df, g1f, g1b, T, B * H)       # d f(t) = G1_fwd(t) + G1_bwd(T-1-t)
                    ops.add_time_reversed(dr, g2b, g2f, T, B * H)       # d b(k) = G2_bwd(k) + G2_fwd(T-1-k)
                else:
                    df.copy_(g1f)
                    ops.add_time_reversed(dr, None, g2f, T, B * H)

    def _aux_forward(self, a, B, Breal, tg, want_probs):
        """style classifier on a decoder head's OUTPUT (reference vae_definition.py:747-761): x*W + b from the (T*B, N) probabilities,
        the recurrence, Dense softmax + loss on the last state"""
        s, P = self.spec, self.P
        r, h, H = a.rec, a.head, s.H
        R = r.T * B
        probs = self._v("out.%s_p" % a.src, R, r.K)
        ops.gemm(probs, P[r.prefix + ".W"], self._v(r.prefix + ".xp", R, s.GH), R, s.GH, r.K, bias=P[r.prefix + ".b"], c_layout=self.lay)
        self._rec_forward(r, B)
        top = self._v(r.prefix + ".hs", r.T + 1, B, H)[r.T]
        ops.head(0, self.kind, B, H, s.C, top, self._v(a.key + ".wt", h.NP, H), P[h.out + ".b"],
                 target_idx=self._v("in.c_idx", B) if tg else None, row_weight=self._v("in.rw_" + a.key, B) if tg else None,
                 grad_scale=a.weight, probs=self._v("out.%s_p" % a.key, B, s.C) if want_probs else None,
                 argmax=self._v(a.key + ".argmax", B), dlogits=self._v(a.key + ".dl", B, h.NP) if (self.training and tg) else None,
                 scalars=self.scal[a.slot:a.slot + 2], b_stride=B, b_valid=Breal)

    def _aux_backward(self, a, B):
        """... and back: Dense, BPTT, the classifier's parameters, then its gradient w.r.t. the source head's PROBABILITIES folded
        into that head's d(logits) (softmax Jacobian) - before the head's own backward pass runs"""
        s, P, G = self.spec, self.P, self.G
        r, h, H = a.rec, a.head, s.H
        R = r.T * B
        dl = self._v(a.key + ".dl", B, h.NP)
        top = self._v(r.prefix + ".hs", r.T + 1, B, H)[r.T]
        dh = self._v(a.key + ".dh", B, H)
        ops.gemm(dl, self._v(a.key + ".wt", h.NP, H), dh, B, H, h.NP)
        self._side(lambda: (ops.gemm(top, dl, G[h.out + ".W"], H, s.C, B, trans_a=True, ldb=h.NP, accumulate=True),
                            ops.colsum(dl, B, s.C, G[h.out + ".b"], ldx=h.NP)))
        self._stack_backward([r], B, dh_last=dh, dh_last_ld=H)
        da = self._v(r.prefix + ".da", R, s.GH)
        probs = self._v("out.%s_p" % a.src, R, r.K)
        self._side(lambda: ops.gemm(probs, da, G[r.prefix + ".W"], r.K, s.GH, R, trans_a=True, accumulate=True,
                                    split_k=self._split_k(R)))
        dp = self._v(a.key + ".dp", R, r.K)
        ops.gemm(da, P[r.prefix + ".W"], dp, R, r.K, s.GH, trans_b=True)
        src = self.head[a.src]
        ops.softmax_bwd_add(probs, dp, self._v(a.src + ".dl", R, src.NP), R, src.N, src.NP)

    def _latent_forward_unfused(self, Breal, B):
        """encoder tail Denses, z_mean / z_log_var, KL + sampling + style softmax, one launch per operation"""
        s, P = self.spec, self.P
        H, Z = s.H, s.Z
        h = self._v("cat", B, self.ncat * H)
        if self.has_pack:
            pk = self._v("pack", B, H)
            ops.gemm(h, P["enc.pack.W"], pk, B, H, self.ncat * H, bias=P["enc.pack.b"], act=hl.ACT_TANH)
            h = pk
        if s.extra_layer:
            ex = self._v("extra", B, H)
            ops.gemm(h, P["enc.extra.W"], ex, B, H, s.tail_in, bias=P["enc.extra.b"], act=hl.ACT_TANH)
            h = ex
        self._tail = h
        h1w = H // 2 if s.split else H
        h2 = h[:, h1w:] if s.split else h
        mu, lv = self._v("mu", B, Z), self._v("lv", B, Z)
        ops.gemm(h, P["enc.zmean.W"], mu, B, Z, h1w, lda=H, bias=P["enc.zmean.b"])
        ops.gemm(h2, P["enc.zlogvar.W"], lv, B, Z, H - h1w if s.split else H, lda=H, bias=P["enc.zlogvar.b"])
        zh = self._v("zh", B, s.zin)
        ops.latent_fwd(Breal, Z, s.C if s.style else 0, s.beta, s.prior_mean, s.prior_std, 1.0 / self.norm_B, mu, lv,
                       self._v("in.eps", B, Z), zh, self.scal[S_KL:S_KL + 3],
                       style_target=self._v("in.c_idx", Breal) if (s.style and self._have_targets) else None,
                       style_row_weight=self._v("in.rw_style", Breal) if (s.style and self._have_targets) else None,
                       style_probs=self._v("style_p", B, s.C) if s.style else None, ldz=s.zin)

    def _latent_backward_unfused(self, Breal, B):
        """initial-state Denses, latent block and encoder tail Denses backward, one launch per operation; returns d(cat)"""
        s, P, G = self.spec, self.P, self.G
        H, Z = s.H, s.Z
        dS = self._v("dS", B, self.n_init * H)
        ldS = self.n_init * H
        # initial-state Denses: S = tanh([z|hist] Winit + b)
        S, zh = self._v("S", B, ldS), self._v("zh", B, s.zin)
        ops.tanh_bwd(S, dS, dS)
        self._side(lambda: (ops.gemm(zh, dS, G["dec.init.W"], s.zin, ldS, B, trans_a=True, accumulate=True),
                            ops.colsum(dS, B, ldS, G["dec.init.b"])))
        dzh = self._v("dzh", B, s.zin)
        ops.gemm(dS, P["dec.init.W"], dzh, B, s.zin, ldS, trans_b=True)
        if s.signature:
            ops.signature_head_bwd(dzh, s.sig_off, s.SD, Breal, self._v("sig.out", B, s.SD), self._v("in.sig", B, s.SD),
                                   self._v("in.rw_sig", B), s.w_sig)
        # ---- latent ------------------------------------------------------------------------------------
        mu, lv = self._v("mu", B, Z), self._v("lv", B, Z)
        dmu, dlv = self._v("dmu", B, Z), self._v("dlv", B, Z)
        if B > Breal:            # padding rows carry no gradient
            dmu[Breal:].zero_()
            dlv[Breal:].zero_()
        ops.latent_bwd(Breal, Z, s.C if s.style else 0, s.beta, s.prior_mean, s.prior_std, s.w_style, 1.0 / self.norm_B, mu, lv,
                       self._v("in.eps", B, Z), dzh, dmu, dlv, style_probs=self._v("style_p", B, s.C) if s.style else None,
                       style_target=self._v("in.c_idx", Breal) if s.style else None,
                       style_row_weight=self._v("in.rw_style", Breal) if s.style else None, lddz=s.zin)
        h = self._tail
        h1w = H // 2 if s.split else H
        h2w = H - h1w if s.split else H
        dt = self._v("dtail", B, H)
        self._side(lambda: (ops.gemm(h, dmu, G["enc.zmean.W"], h1w, Z, B, trans_a=True, lda=H, accumulate=True),
                            ops.colsum(dmu, B, Z, G["enc.zmean.b"]),
                            ops.gemm(h[:, h1w:] if s.split else h, dlv, G["enc.zlogvar.W"], h2w, Z, B, trans_a=True, lda=H,
                                     accumulate=True),
                            ops.colsum(dlv, B, Z, G["enc.zlogvar.b"])))
        if s.split:
            ops.gemm(dmu, P["enc.zmean.W"], dt, B, h1w, Z, trans_b=True, ldc=H)
            ops.gemm(dlv, P["enc.zlogvar.W"], dt[:, h1w:], B, h2w, Z, trans_b=True, ldc=H)
        else:
            ops.gemm(dmu, P["enc.zmean.W"], dt, B, H, Z, trans_b=True)
            dt2 = self._v("dtail2", B, H)
            ops.gemm(dlv, P["enc.zlogvar.W"], dt2, B, H, Z, trans_b=True)
            dt.add_(dt2)
        # ---- encoder tail ------------------------------------------------------------------------------
        if s.extra_layer:
            ex = self._v("extra", B, H)
            src = self._v("pack", B, H) if self.has_pack else self._v("cat", B, self.ncat * H)
            ops.tanh_bwd(ex, dt, dt)
            self._side(lambda dt=dt: (ops.gemm(src, dt, G["enc.extra.W"], s.tail_in, H, B, trans_a=True, accumulate=True),
                                      ops.colsum(dt, B, H, G["enc.extra.b"])))
            dt2 = self._v("dcat", B, s.tail_in) if not self.has_pack else self._v("dtail2", B, H)
            ops.gemm(dt, P["enc.extra.W"], dt2, B, s.tail_in, H, trans_b=True)
            dt = dt2
        ldc = self.ncat * H
        if self.has_pack:
            pk, cat = self._v("pack", B, H), self._v("cat", B, ldc)
            ops.tanh_bwd(pk, dt, dt)
            self._side(lambda dt=dt: (ops.gemm(cat, dt, G["enc.pack.W"], ldc, H, B, trans_a=True, accumulate=True),
                                      ops.colsum(dt, B, H, G["enc.pack.b"])))
            dcat = self._v("dcat", B, ldc)
            ops.gemm(dt, P["enc.pack.W"], dcat, B, ldc, H, trans_b=True)
        else:
            dcat = dt
        return dcat

    def _signature_forward(self, Breal, B):
        """signature head (reference vae_definition.py:737-745): tanh of the latent columns behind the style classifier's"""
        s = self.spec
        if not s.signature:
            return
        tg = self._have_targets
        ops.signature_head_fwd(self._v("zh", B, s.zin), s.sig_off, s.SD, Breal, self._v("sig.out", B, s.SD),
                               target=self._v("in.sig", B, s.SD) if tg else None,
                               row_weight=self._v("in.rw_sig", B) if tg else None, scalars=self.scal[S_SIG_LOSS:S_SIG_LOSS + 2])
