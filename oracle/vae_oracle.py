"""CPU oracle of the MIDI-VAE train / inference step  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (midi-vae_amd/) never calls it and fails loudly without its HIP library.

PARITY STATUS: **parity unpinned** for the neural arithmetic.  The reference builds its graph from
Keras (~2.0.8) layers and recurrentshop cells; neither dependency is vendored under /root/reference,
neither is installable here (no network), and the reference ships no tests, weights or recorded losses
(SURVEY.md F2/F3/F5/F7).  This file therefore RESTATES the published semantics of those layers
 in float64 NumPy, anchored on the reference's own call sites, cited inline.
What IS pinned to the reference: the packers / argmax decode around this graph (tests/golden, captured
from the reference's NumPy helpers), and closed-form known answers (tests/test_oracle_known_answers.py).
Gradients are pinned by central finite differences and by an independent torch-autograd restatement.

Graph (reference vae_definition.py):
  encoder  :443-516   stacked RNN over notes (:455-461), one RNN per meta roll (:464-480), concat,
                      Dense+tanh pack (:483-484), Dense+tanh extra (:486-487), split halves (:489-492),
                      z_mean / z_log_var Dense (:506-507), KL layer (:15-37,:514), sampling (:498-502,:515)
  decoder  :519-645   per head: initial state = Dense(tanh)([z, history]) (:548-568), recurrentshop cell
                      stack stepped output_length times on a CONSTANT input (F9 / Appendix A.6),
                      Dense(activation) on the top cell's output (:542,:593,:631)
  style    :730-734   softmax over z[:, :num_composers]
  losses   :332-441   Keras weighted losses + KL; optimizer :174-175 (Keras Adam, Appendix A.8)

Conventions: gate blocks along the last axis are [z|r|h] (GRU), [i|f|g|o] (LSTM), [h] (SimpleRNN);
W (in, G*H), U (H, G*H), b (G*H,).  Sequences are time-major (T, B, .) inside this file.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

GATES = {"GRU": 3, "LSTM": 4, "SimpleRNN": 1}
NSTATE = {"GRU": 1, "LSTM": 2, "SimpleRNN": 1}
CE_EPS = 1e-7  # Keras backend epsilon used by categorical_crossentropy's clip


# -------------------------------------------------------------------------------------------------
# configuration
# -------------------------------------------------------------------------------------------------

DEFAULT_CFG = dict(
    cell="GRU", H=256, Z=256, Din=61, Dout=61, T=64, V=4, ID=16, C=2,
    Le=2, Ld=2, meta_instrument=True, meta_velocity=True, extra_layer=True, split=True, history=True,
    style=True, w_instr=0.1, w_vel=1.0, w_style=0.1, beta=0.1, prior_mean=0.0, prior_std=1.0,
    lr=2e-4, optimizer="Adam",
    meta_held=False, w_held=1.0, meta_next=False, w_next=1.0,      # reference settings.py:217,227 (off by default)
    signature=False, SD=15, w_sig=1.0,                             # reference settings.py:189-192
    comp_notes=False, w_cnotes=1.0, comp_instr=False, w_cinstr=1.0,   # reference settings.py:195-200
    add_dim=0,                                                      # decoder_additional_input_dim, settings.py:167-177
    bidirectional=False,                                            # settings.py:159
    # SURVEY A.6 / F9 hedge, ORACLE ONLY, forward only, excluded from every parity claim: "none" = the decoder as written (the
    # readout is accepted and dropped: x_t = start); "add" = the other reading of recurrentshop's readout, x_t = start + y_{t-1}
    # with y_{-1} = initial_readout = start.  The product implements "none"; "add" exists so that its cost can be evaluated.
    readout="none",
)


def enc_notes_layers(cfg):
    """[(prefix, reversed?, input width)] of the encoder's notes stack, bottom to top.  Unidirectional: Le layers.  Bidirectional
    (reference vae_definition.py:445-453, as written): ``range(1, Le-1)`` builds Le-2 Bidirectional(..., merge_mode='concat')
    layers - each a forward and a backward RNN whose outputs are concatenated per time step - and ONE unidirectional layer on top;
    with the default Le=2 that is no bidirectional layer at all."""
    H = cfg["H"]
    if not cfg["bidirectional"]:
        return [[("enc.notes.%d" % l, False, cfg["Din"] if l == 0 else H)] for l in range(cfg["Le"])]
    nbi = max(cfg["Le"] - 2, 0)
    out = []
    for l in range(nbi):
        k = cfg["Din"] if l == 0 else 2 * H
        out.append([("enc.notes.%d" % l, False, k), ("enc.notes.%d.rev" % l, True, k)])
    out.append([("enc.notes.%d" % nbi, False, cfg["Din"] if nbi == 0 else 2 * H)])
    return out


def make_cfg(**kw):
    cfg = dict(DEFAULT_CFG)
    unknown = set(kw) - set(cfg)
    if unknown:
        raise KeyError(sorted(unknown))
    cfg.update(kw)
    return cfg


def param_shapes(cfg):
    """Ordered name -> shape of every trainable tensor (the naming contract shared with the product's
    layout.py; a test checks both agree)."""
    G, H, Z = GATES[cfg["cell"]], cfg["H"], cfg["Z"]
    ns = NSTATE[cfg["cell"]]
    GH = G * H
    P = OrderedDict()

    def rnn(prefix, k):
        P[prefix + ".W"] = (k, GH)
        P[prefix + ".U"] = (H, GH)
        P[prefix + ".b"] = (GH,)

    for layer in enc_notes_layers(cfg):
        for prefix, _, k in layer:
            rnn(prefix, k)
    ncat = 1
    if cfg["meta_instrument"]:
        rnn("enc.instr", cfg["ID"])
        ncat += 1
    if cfg["meta_velocity"]:
        rnn("enc.vel", 1)
        ncat += 1
    if cfg["meta_held"]:                                    # reference :476-480: RNN over the (T,2) held-notes roll
        rnn("enc.held", 2)
        ncat += 1
    packed = cfg["meta_instrument"] or cfg["meta_velocity"]   # reference :483 (condition as written: held alone does not pack)
    if packed:
        P["enc.pack.W"], P["enc.pack.b"] = (ncat * H, H), (H,)
    if cfg["extra_layer"]:
        P["enc.extra.W"], P["enc.extra.b"] = (H if packed else ncat * H, H), (H,)
    h1 = H // 2 if cfg["split"] else H
    h2 = H - H // 2 if cfg["split"] else H
    P["enc.zmean.W"], P["enc.zmean.b"] = (h1, Z), (Z,)
    P["enc.zlogvar.W"], P["enc.zlogvar.b"] = (h2, Z), (Z,)
    zin = (2 * Z if cfg["history"] else Z) + cfg["add_dim"]

    def init(prefix):
        for s in range(ns):
            P["%s.%d.W" % (prefix, s)], P["%s.%d.b" % (prefix, s)] = (zin, H), (H,)

    for l in range(cfg["Ld"]):
        init("dec.notes.init.%d" % l)
    for l in range(cfg["Ld"]):
        rnn("dec.notes.%d" % l, cfg["Dout"] if l == 0 else H)
    P["dec.notes.out.W"], P["dec.notes.out.b"] = (H, cfg["Dout"]), (cfg["Dout"],)
    if cfg["meta_instrument"]:
        init("dec.instr.init")
        rnn("dec.instr.cell", cfg["ID"])
        P["dec.instr.out.W"], P["dec.instr.out.b"] = (H, cfg["ID"]), (cfg["ID"],)
    if cfg["meta_velocity"]:
        init("dec.vel.init")
        rnn("dec.vel.cell", 1)
        P["dec.vel.out.W"], P["dec.vel.out.b"] = (H, 1), (1,)
    if cfg["meta_held"]:                                    # reference :648-683: one cell, Dense(2, softmax)
        init("dec.held.init")
        rnn("dec.held.cell", 2)
        P["dec.held.out.W"], P["dec.held.out.b"] = (H, 2), (2,)
    if cfg["meta_next"]:                                    # reference :685-726: a second Ld-layer stack, Dense(D, softmax)
        for l in range(cfg["Ld"]):
            init("dec.next.init.%d" % l)
        for l in range(cfg["Ld"]):
            rnn("dec.next.%d" % l, cfg["Dout"] if l == 0 else H)
        P["dec.next.out.W"], P["dec.next.out.b"] = (H, cfg["Dout"]), (cfg["Dout"],)
    if cfg["comp_notes"]:                                   # reference :747-753: Keras RNN over the notes OUTPUT -> Dense(C)
        rnn("cnotes.rnn", cfg["Dout"])
        P["cnotes.out.W"], P["cnotes.out.b"] = (H, cfg["C"]), (cfg["C"],)
    if cfg["comp_instr"]:                                   # reference :755-761: the same over the instrument OUTPUT
        rnn("cinstr.rnn", cfg["ID"])
        P["cinstr.out.W"], P["cinstr.out.b"] = (H, cfg["C"]), (cfg["C"],)
    return P


# -------------------------------------------------------------------------------------------------
# primitives
# -------------------------------------------------------------------------------------------------

def hard_sigmoid(x):
    """Keras hard_sigmoid: clip(0.2 x + 0.5, 0, 1) (Appendix A.2)."""
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


def _dhs(y):
    """d hard_sigmoid / dx from its OUTPUT: 0.2 strictly inside (0,1), else 0."""
    return 0.2 * ((y > 0.0) & (y < 1.0))


def softmax(x):
    e = np.exp(x - np.max(x, axis=-1, keepdims=True))
    return e / np.sum(e, axis=-1, keepdims=True)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# recurrent activations by name: (function, derivative from the OUTPUT).  "hard_sigmoid" is what the reference's layers use
# (Keras / recurrentshop defaults); "sigmoid" exists for tests/test_oracle_third_party_cpu.py ONLY:
# with it the cells below are the cells torch.nn.LSTMCell / GRUCell implement, an implementation this repo did not write.
_REC_ACT = {"hard_sigmoid": (hard_sigmoid, _dhs), "sigmoid": (sigmoid, lambda y: y * (1.0 - y))}


def rnn_forward(cell, xp, U, h0, c0=None, rec_act="hard_sigmoid"):
    """Recurrence over pre-projected inputs.  xp (T,B,G*H) already holds x_t W + b.
    Returns hs (T+1,B,H) with hs[0]=h0, cs (T+1,B,H) (LSTM else None), acts (T,B,G*H) = post-activation gates.
    GRU uses the Keras-2.0.x 'reset before matmul' form (Appendix A.2 / A.5).  ``rec_act``: test-only switch (_REC_ACT)."""
    hard_sigmoid = _REC_ACT[rec_act][0]          # (shadows the module function inside this call)
    T, B, GH = xp.shape
    H = U.shape[0]
    hs = np.zeros((T + 1, B, H), xp.dtype)
    hs[0] = h0
    acts = np.zeros((T, B, GH), xp.dtype)
    cs = None
    if cell == "LSTM":
        cs = np.zeros((T + 1, B, H), xp.dtype)
        cs[0] = c0
    for t in range(T):
        h = hs[t]
        if cell == "GRU":
            a = xp[t, :, :2 * H] + h @ U[:, :2 * H]
            z, r = hard_sigmoid(a[:, :H]), hard_sigmoid(a[:, H:])
            hh = np.tanh(xp[t, :, 2 * H:] + (r * h) @ U[:, 2 * H:])
            hs[t + 1] = z * h + (1.0 - z) * hh
            acts[t, :, :H], acts[t, :, H:2 * H], acts[t, :, 2 * H:] = z, r, hh
        elif cell == "LSTM":
            a = xp[t] + h @ U
            i, f = hard_sigmoid(a[:, :H]), hard_sigmoid(a[:, H:2 * H])
            g, o = np.tanh(a[:, 2 * H:3 * H]), hard_sigmoid(a[:, 3 * H:])
            cs[t + 1] = f * cs[t] + i * g
            hs[t + 1] = o * np.tanh(cs[t + 1])
            acts[t, :, :H], acts[t, :, H:2 * H], acts[t, :, 2 * H:3 * H], acts[t, :, 3 * H:] = i, f, g, o
        else:
            hs[t + 1] = np.tanh(xp[t] + h @ U)
            acts[t] = hs[t + 1]
    return hs, cs, acts


def rnn_backward(cell, hs, cs, acts, U, dhs_ext=None, dh_last=None, rec_act="hard_sigmoid"):
    """BPTT.  dhs_ext (T,B,H): gradient arriving at every h_t (t=1..T stored at index t-1) from above;
    dh_last (B,H): extra gradient at the final state.  Returns da (T,B,G*H) (= d xp), dU, dh0, dc0."""
    _dhs = _REC_ACT[rec_act][1]                  # (test-only switch, see rnn_forward)
    T, B, GH = acts.shape
    H = U.shape[0]
    da = np.zeros_like(acts)
    dU = np.zeros_like(U)
    dh = np.zeros((B, H), acts.dtype) if dh_last is None else dh_last.copy()
    dc = np.zeros((B, H), acts.dtype)
    for t in range(T - 1, -1, -1):
        d = dh + (dhs_ext[t] if dhs_ext is not None else 0.0)
        hp = hs[t]
        if cell == "GRU":
            z, r, hh = acts[t, :, :H], acts[t, :, H:2 * H], acts[t, :, 2 * H:]
            da_h = d * (1.0 - z) * (1.0 - hh * hh)
            drh = da_h @ U[:, 2 * H:].T
            da_z = d * (hp - hh) * _dhs(z)
            da_r = drh * hp * _dhs(r)
            da[t, :, :H], da[t, :, H:2 * H], da[t, :, 2 * H:] = da_z, da_r, da_h
            dU[:, :2 * H] += hp.T @ da[t, :, :2 * H]
            dU[:, 2 * H:] += (r * hp).T @ da_h
            dh = d * z + drh * r + da[t, :, :2 * H] @ U[:, :2 * H].T
        elif cell == "LSTM":
            i, f, g, o = (acts[t, :, :H], acts[t, :, H:2 * H], acts[t, :, 2 * H:3 * H], acts[t, :, 3 * H:])
            tc = np.tanh(cs[t + 1])
            dct = dc + d * o * (1.0 - tc * tc)
            da[t, :, :H] = dct * g * _dhs(i)
            da[t, :, H:2 * H] = dct * cs[t] * _dhs(f)
            da[t, :, 2 * H:3 * H] = dct * i * (1.0 - g * g)
            da[t, :, 3 * H:] = d * tc * _dhs(o)
            dc = dct * f
            dU += hp.T @ da[t]
            dh = da[t] @ U.T
        else:
            y = acts[t]
            da[t] = d * (1.0 - y * y)
            dU += hp.T @ da[t]
            dh = da[t] @ U.T
    return da, dU, dh, dc


def _cce(p, y):
    """Keras categorical_crossentropy on probabilities: renormalise, clip to [eps,1-eps], -sum y log p."""
    q = p / np.sum(p, axis=-1, keepdims=True)
    q = np.clip(q, CE_EPS, 1.0 - CE_EPS)
    return -np.sum(y * np.log(q), axis=-1)


def _cce_grad_logits(p, y):
    """d CE / d logits for softmax -> CE with the clip's pass-through gradient: (p*sum(y) - y) where the target
    probabilities lie inside the clip range; entries whose p is outside the range contribute no gradient."""
    inside = (p >= CE_EPS) & (p <= 1.0 - CE_EPS)
    ym = y * inside
    return p * np.sum(ym, axis=-1, keepdims=True) - ym


def _weighted_mean(score, w):
    """Keras 2.0.8 weighted objective: mean(score * w) / mean(w != 0).  Returns (loss, dloss/dscore)."""
    nz = np.mean((w != 0).astype(score.dtype))
    return np.mean(score * w) / nz, w / (nz * score.size)


# -------------------------------------------------------------------------------------------------
# the model
# -------------------------------------------------------------------------------------------------

class OracleVAE(object):
    def __init__(self, cfg, dtype=np.float64):
        self.cfg = dict(cfg)
        self.dtype = dtype
        self.shapes = param_shapes(cfg)

    # ---- helpers --------------------------------------------------------------------------------
    def _states(self, p, prefix, zh, cache, key):
        """Initial states of one decoder cell: tanh(Dense([z, history])) per state (reference :558-568)."""
        out = []
        for s in range(NSTATE[self.cfg["cell"]]):
            out.append(np.tanh(zh @ p["%s.%d.W" % (prefix, s)] + p["%s.%d.b" % (prefix, s)]))
        cache[key] = out
        return out

    def _enc_rnn(self, p, prefix, x_tm):
        """x_tm (T,B,K) -> final h, cache."""
        cell, H = self.cfg["cell"], self.cfg["H"]
        B = x_tm.shape[1]
        xp = x_tm @ p[prefix + ".W"] + p[prefix + ".b"]
        z0 = np.zeros((B, H), self.dtype)
        hs, cs, acts = rnn_forward(cell, xp, p[prefix + ".U"], z0, z0 if cell == "LSTM" else None)
        return hs, cs, acts

    # ---- encoder --------------------------------------------------------------------------------
    def encode(self, p, X, I=None, Vel=None, eps=None, cache=None, Held=None):
        """X (B,T,Din), I (B,V,ID), Vel (B,T,1), Held (B,T,2) one-hot, eps (B,Z) ALREADY scaled by epsilon_std.
        Returns z (B,Z); fills cache with mu, logvar and everything backward needs."""
        cfg, dt = self.cfg, self.dtype
        c = {} if cache is None else cache
        x = np.asarray(X, dt).transpose(1, 0, 2)
        seqs = []
        for layer in enc_notes_layers(cfg):
            outs = []
            for prefix, rev, _ in layer:
                xin = x[::-1] if rev else x              # Keras Bidirectional: the backward layer reads the sequence reversed ...
                hs, cs, acts = self._enc_rnn(p, prefix, xin)
                seqs.append((prefix, rev, xin, hs, cs, acts))
                outs.append(hs[1:][::-1] if rev else hs[1:])     # ... and its outputs are put back in time order
            x = outs[0] if len(outs) == 1 else np.concatenate(outs, axis=-1)
        c["enc_notes"] = seqs
        feats = [seqs[-1][3][-1]]
        if cfg["meta_instrument"]:
            xi = np.asarray(I, dt).transpose(1, 0, 2)
            hs, cs, acts = self._enc_rnn(p, "enc.instr", xi)
            c["enc_instr"] = (xi, hs, cs, acts)
            feats.append(hs[-1])
        if cfg["meta_velocity"]:
            xv = np.asarray(Vel, dt).transpose(1, 0, 2)
            hs, cs, acts = self._enc_rnn(p, "enc.vel", xv)
            c["enc_vel"] = (xv, hs, cs, acts)
            feats.append(hs[-1])
        if cfg["meta_held"]:
            xd = np.asarray(Held, dt).transpose(1, 0, 2)
            hs, cs, acts = self._enc_rnn(p, "enc.held", xd)
            c["enc_held"] = (xd, hs, cs, acts)
            feats.append(hs[-1])
        h = np.concatenate(feats, axis=1)
        c["cat"] = h
        if "enc.pack.W" in p:
            h = np.tanh(h @ p["enc.pack.W"] + p["enc.pack.b"])
            c["pack"] = h
        if cfg["extra_layer"]:
            c["extra_in"] = h
            h = np.tanh(h @ p["enc.extra.W"] + p["enc.extra.b"])
            c["extra"] = h
        H = cfg["H"]
        h1, h2 = (h[:, :H // 2], h[:, H // 2:]) if cfg["split"] else (h, h)
        c["h1"], c["h2"] = h1, h2
        mu = h1 @ p["enc.zmean.W"] + p["enc.zmean.b"]
        lv = h2 @ p["enc.zlogvar.W"] + p["enc.zlogvar.b"]
        if eps is None:
            eps = np.zeros_like(mu)
        z = mu + np.exp(lv / 2.0) * eps
        c["mu"], c["lv"], c["eps"], c["z"] = mu, lv, np.asarray(eps, dt), z
        return z

    # ---- decoder --------------------------------------------------------------------------------
    def _dec_head(self, p, cells, inits, outprefix, start, zh, steps, cache, key, act=None):
        """One decoder head: ``cells`` / ``inits`` are the per-layer parameter prefixes of the cell stack and
        of its initial-state Denses.  ``act``: the head's output activation (readout="add" only)."""
        cell = self.cfg["cell"]
        if self.cfg.get("readout", "none") == "add":
            return self._dec_head_readout_add(p, cells, inits, outprefix, start, zh, steps, cache, key, act)
        layers = []
        x_seq = None
        for l, (cp, ip) in enumerate(zip(cells, inits)):
            st = self._states(p, ip, zh, cache, key + ".init%d" % l)
            if l == 0:
                xp0 = start @ p[cp + ".W"] + p[cp + ".b"]                      # constant input, Appendix A.6
                xp = np.broadcast_to(xp0[None], (steps,) + xp0.shape).copy()
            else:
                xp = x_seq @ p[cp + ".W"] + p[cp + ".b"]
            hs, cs, acts = rnn_forward(cell, xp, p[cp + ".U"], st[0], st[1] if cell == "LSTM" else None)
            layers.append((cp, ip, x_seq, hs, cs, acts))
            x_seq = hs[1:]
        cache[key] = layers
        return x_seq @ p[outprefix + ".W"] + p[outprefix + ".b"]           # (steps,B,out) logits

    def _dec_head_readout_add(self, p, cells, inits, outprefix, start, zh, steps, cache, key, act):
        """SURVEY A.6, the alternative reading: the previous step's OUTPUT is fed back, x_t = start + y_{t-1} (y_{-1} = start).
        Step by step through the whole stack; no backward pass exists for it (OracleVAE.backward raises)."""
        cell = self.cfg["cell"]
        st = [self._states(p, ip, zh, cache, key + ".init%d" % l) for l, ip in enumerate(inits)]
        h = [s_[0] for s_ in st]
        c = [s_[1] if cell == "LSTM" else None for s_ in st]
        y, logits = start, []
        for _ in range(steps):
            x = start + y
            for l, cp in enumerate(cells):
                hs, cs, _ = rnn_forward(cell, (x @ p[cp + ".W"] + p[cp + ".b"])[None], p[cp + ".U"], h[l], c[l])
                h[l], c[l] = hs[1], (cs[1] if cs is not None else None)
                x = hs[1]
            lg = x @ p[outprefix + ".W"] + p[outprefix + ".b"]
            logits.append(lg)
            y = act(lg)
        cache[key] = None
        return np.stack(logits)

    def decode(self, p, z, hist, starts, cache=None, add=None):
        """z (B,Z), hist (B,Z) or None, starts = dict(notes (B,Dout), instr (B,ID), vel (B,)), add (B,add_dim) = the decoder's
        additional input (reference :553-556).  Returns dict of batch-major outputs: notes (B,T,Dout) probs, instr (B,V,ID)
        probs, vel (B,T,1)."""
        cfg, dt = self.cfg, self.dtype
        c = {} if cache is None else cache
        zh = np.concatenate([z, np.asarray(hist, dt)], axis=1) if cfg["history"] else z
        if cfg["add_dim"]:
            zh = np.concatenate([zh, np.asarray(add, dt)], axis=1)
        c["zh"] = zh
        out = {}
        Ld = cfg["Ld"]
        lg = self._dec_head(p, ["dec.notes.%d" % l for l in range(Ld)], ["dec.notes.init.%d" % l for l in range(Ld)],
                            "dec.notes.out", np.asarray(starts["notes"], dt), zh, cfg["T"], c, "dec_notes", act=softmax)
        out["notes"] = softmax(lg).transpose(1, 0, 2)
        if cfg["meta_instrument"]:
            lg = self._dec_head(p, ["dec.instr.cell"], ["dec.instr.init"], "dec.instr.out",
                                np.asarray(starts["instr"], dt), zh, cfg["V"], c, "dec_instr", act=softmax)
            out["instr"] = softmax(lg).transpose(1, 0, 2)
        if cfg["meta_velocity"]:
            lg = self._dec_head(p, ["dec.vel.cell"], ["dec.vel.init"], "dec.vel.out",
                                np.asarray(starts["vel"], dt).reshape(-1, 1), zh, cfg["T"], c, "dec_vel", act=sigmoid)
            out["vel"] = sigmoid(lg).transpose(1, 0, 2)
        if cfg["meta_held"]:
            lg = self._dec_head(p, ["dec.held.cell"], ["dec.held.init"], "dec.held.out",
                                np.asarray(starts.get("held", np.zeros((z.shape[0], 2))), dt), zh, cfg["T"], c, "dec_held", act=softmax)
            out["held"] = softmax(lg).transpose(1, 0, 2)
        if cfg["meta_next"]:
            lg = self._dec_head(p, ["dec.next.%d" % l for l in range(Ld)], ["dec.next.init.%d" % l for l in range(Ld)],
                                "dec.next.out", np.asarray(starts.get("next", np.zeros((z.shape[0], cfg["Dout"]))), dt), zh,
                                cfg["T"], c, "dec_next", act=softmax)
            out["next"] = softmax(lg).transpose(1, 0, 2)
        return out

    # ---- full forward with losses ---------------------------------------------------------------
    def forward(self, p, batch, eps):
        """batch: X, I, Vel, Hist, Y, C (B,Cn one-hot), w_notes (B,T), w_instr/w_vel/w_style (B,), and optional
        start_notes/start_instr/start_vel (default zeros).  eps (B,Z) already scaled by epsilon_std.
        Returns (metrics dict, cache)."""
        cfg, dt = self.cfg, self.dtype
        c = {}
        B = np.asarray(batch["X"]).shape[0]
        z = self.encode(p, batch["X"], batch.get("I"), batch.get("Vel"), eps, c, Held=batch.get("Held"))
        starts = dict(notes=batch.get("start_notes", np.zeros((B, cfg["Dout"]))),
                      instr=batch.get("start_instr", np.zeros((B, cfg["ID"]))),
                      vel=batch.get("start_vel", np.zeros((B,))),
                      held=batch.get("start_held", np.zeros((B, 2))),
                      next=batch.get("start_next", np.zeros((B, cfg["Dout"]))))
        c["starts"] = starts
        out = self.decode(p, z, batch.get("Hist", np.zeros((B, cfg["Z"]))), starts, c, add=batch.get("Add"))
        c["out"] = out
        m = OrderedDict()
        ones = np.ones((B,), dt)
        # KL (reference :29-37)
        mu, lv = c["mu"], c["lv"]
        pm, ps = cfg["prior_mean"], cfg["prior_std"]
        plv, pvar = 2.0 * np.log(ps), ps * ps
        kl_b = cfg["beta"] * (-0.5 * np.sum(1.0 + lv - plv - ((mu - pm) ** 2 + np.exp(lv)) / pvar, axis=1))
        m["kl"] = np.mean(kl_b)
        # notes: temporal sample weights (reference :336-338)
        Y = np.asarray(batch["Y"], dt)
        wn = np.asarray(batch.get("w_notes", np.ones(Y.shape[:2])), dt)
        sc = _cce(out["notes"], Y)
        m["notes_loss"], c["g_notes"] = _weighted_mean(sc, wn)
        m["notes_acc"] = np.mean(np.argmax(out["notes"], -1) == np.argmax(Y, -1))
        total = m["notes_loss"] + m["kl"]
        if cfg["meta_instrument"]:
            It = np.asarray(batch["I"], dt)
            wi = np.asarray(batch.get("w_instr", ones), dt)
            sc = np.mean(_cce(out["instr"], It), axis=1)
            m["instr_loss"], g = _weighted_mean(sc, wi)
            c["g_instr"] = g[:, None] / cfg["V"] * np.ones((1, cfg["V"]))
            m["instr_acc"] = np.mean(np.argmax(out["instr"], -1) == np.argmax(It, -1))
            total = total + cfg["w_instr"] * m["instr_loss"]
        if cfg["meta_velocity"]:
            Vt = np.asarray(batch["Vel"], dt)
            wv = np.asarray(batch.get("w_vel", ones), dt)
            sc = np.mean((out["vel"][..., 0] - Vt[..., 0]) ** 2, axis=1)
            m["vel_loss"], g = _weighted_mean(sc, wv)
            c["g_vel"] = g[:, None] / cfg["T"] * np.ones((1, cfg["T"]))
            m["vel_acc"] = np.mean(np.round(out["vel"]) == Vt)      # Keras binary_accuracy
            total = total + cfg["w_vel"] * m["vel_loss"]
        if cfg["meta_held"]:                                 # target = the held-notes roll itself (reference :1006-1016)
            Dt = np.asarray(batch["Held"], dt)
            wh = np.asarray(batch.get("w_held", ones), dt)
            sc = np.mean(_cce(out["held"], Dt), axis=1)
            m["held_loss"], g = _weighted_mean(sc, wh)
            c["g_held"] = g[:, None] / cfg["T"] * np.ones((1, cfg["T"]))
            m["held_acc"] = np.mean(np.argmax(out["held"], -1) == np.argmax(Dt, -1))
            total = total + cfg["w_held"] * m["held_loss"]
        if cfg["meta_next"]:                                 # target = the NEXT window's notes (reference :882-891,1018-1028)
            Nt = np.asarray(batch["Next"], dt)
            wx = np.asarray(batch.get("w_next", ones), dt)
            sc = np.mean(_cce(out["next"], Nt), axis=1)
            m["next_loss"], g = _weighted_mean(sc, wx)
            c["g_next"] = g[:, None] / cfg["T"] * np.ones((1, cfg["T"]))
            m["next_acc"] = np.mean(np.argmax(out["next"], -1) == np.argmax(Nt, -1))
            total = total + cfg["w_next"] * m["next_loss"]
        if cfg["style"]:
            Ct = np.asarray(batch["C"], dt)
            ws = np.asarray(batch.get("w_style", ones), dt)
            ps_ = softmax(z[:, :cfg["C"]])
            c["p_style"] = ps_
            m["style_loss"], c["g_style"] = _weighted_mean(_cce(ps_, Ct), ws)
            m["style_acc"] = np.mean(np.argmax(ps_, -1) == np.argmax(Ct, -1))
            total = total + cfg["w_style"] * m["style_loss"]
            out["style"] = ps_
        if cfg["signature"]:                                 # reference :737-745: tanh(z[:, off:off+SD]) vs the signature vector, mse
            off = cfg["C"] if cfg["style"] else 0
            St = np.asarray(batch["S"], dt)
            wg = np.asarray(batch.get("w_sig", ones), dt)
            sig = np.tanh(z[:, off:off + cfg["SD"]])
            c["sig"] = sig
            m["sig_loss"], c["g_sig"] = _weighted_mean(np.mean((sig - St) ** 2, axis=1), wg)
            m["sig_acc"] = np.mean(np.argmax(sig, -1) == np.argmax(St, -1))
            total = total + cfg["w_sig"] * m["sig_loss"]
            out["sig"] = sig
        for key, src, flag, w in (("cnotes", "notes", "comp_notes", "w_cnotes"), ("cinstr", "instr", "comp_instr", "w_cinstr")):
            if cfg[flag]:                                    # reference :747-761: Keras RNN over a decoder OUTPUT -> Dense softmax
                Ct = np.asarray(batch["C"], dt)
                x_tm = out[src].transpose(1, 0, 2)
                hs, cs, acts = self._enc_rnn(p, key + ".rnn", x_tm)
                pc = softmax(hs[-1] @ p[key + ".out.W"] + p[key + ".out.b"])
                c[key] = (x_tm, hs, cs, acts, pc)
                wq = np.asarray(batch.get("w_" + key, ones), dt)
                m[key + "_loss"], c["g_" + key] = _weighted_mean(_cce(pc, Ct), wq)
                m[key + "_acc"] = np.mean(np.argmax(pc, -1) == np.argmax(Ct, -1))
                total = total + cfg[w] * m[key + "_loss"]
                out[key] = pc
        m["loss"] = total
        c["batch"] = batch
        return m, c

    # ---- backward -------------------------------------------------------------------------------
    def _dec_head_backward(self, p, g, layers, dlogits, outprefix, start, dzh):
        """dlogits (steps,B,out).  Accumulates parameter grads into g and d[z,history] into dzh."""
        cell = self.cfg["cell"]
        top = layers[-1][3][1:]
        g[outprefix + ".W"] = np.einsum("tbh,tbo->ho", top, dlogits)
        g[outprefix + ".b"] = dlogits.sum((0, 1))
        dh_ext = dlogits @ p[outprefix + ".W"].T
        zh = self._zh
        for l in range(len(layers) - 1, -1, -1):
            cp, ip, x_seq, hs, cs, acts = layers[l]
            da, dU, dh0, dc0 = rnn_backward(cell, hs, cs, acts, p[cp + ".U"], dh_ext)
            g[cp + ".U"] = dU
            g[cp + ".b"] = da.sum((0, 1))
            if l == 0:
                g[cp + ".W"] = start.T @ da.sum(0)
            else:
                g[cp + ".W"] = np.einsum("tbk,tbn->kn", x_seq, da)
                dh_ext = da @ p[cp + ".W"].T
            for s, dst in enumerate([dh0, dc0][:NSTATE[cell]]):
                st = (hs if s == 0 else cs)[0]
                dpre = dst * (1.0 - st * st)
                g["%s.%d.W" % (ip, s)] = zh.T @ dpre
                g["%s.%d.b" % (ip, s)] = dpre.sum(0)
                dzh += dpre @ p["%s.%d.W" % (ip, s)].T

    def _enc_rnn_backward(self, p, g, prefix, rec, dhs_ext, dh_last, need_dx):
        x, hs, cs, acts = rec
        da, dU, _, _ = rnn_backward(self.cfg["cell"], hs, cs, acts, p[prefix + ".U"], dhs_ext, dh_last)
        g[prefix + ".U"] = dU
        g[prefix + ".b"] = da.sum((0, 1))
        g[prefix + ".W"] = np.einsum("tbk,tbn->kn", x, da)
        return da @ p[prefix + ".W"].T if need_dx else None

    def backward(self, p, c):
        """Gradients of metrics['loss'] w.r.t. every parameter (dict name -> array)."""
        cfg, dt = self.cfg, self.dtype
        if cfg.get("readout", "none") != "none":
            raise NotImplementedError("readout=%r is a forward-only study switch of the oracle (SURVEY A.6)" % cfg["readout"])
        b = c["batch"]
        out = c["out"]
        g = {}
        H, Z = cfg["H"], cfg["Z"]
        B = c["mu"].shape[0]
        self._zh = c["zh"]
        dzh = np.zeros_like(c["zh"])
        # classifiers on the decoder's outputs: their gradient arrives at the softmax PROBABILITIES of the notes / instrument heads
        dprob = {}
        for key, src, flag, w in (("cnotes", "notes", "comp_notes", "w_cnotes"), ("cinstr", "instr", "comp_instr", "w_cinstr")):
            if cfg[flag]:
                x_tm, hs, cs, acts, pc = c[key]
                Ct = np.asarray(b["C"], dt)
                dlc = _cce_grad_logits(pc, Ct) * (cfg[w] * c["g_" + key])[:, None]
                g[key + ".out.W"], g[key + ".out.b"] = hs[-1].T @ dlc, dlc.sum(0)
                dx = self._enc_rnn_backward(p, g, key + ".rnn", (x_tm, hs, cs, acts), None, dlc @ p[key + ".out.W"].T, True)
                pr = out[src].transpose(1, 0, 2)
                dprob[src] = pr * (dx - np.sum(pr * dx, axis=-1, keepdims=True))       # softmax Jacobian -> d(logits), time-major
        # heads
        Y = np.asarray(b["Y"], dt)
        dl = (_cce_grad_logits(out["notes"], Y) * c["g_notes"][..., None]).transpose(1, 0, 2)
        if "notes" in dprob:
            dl = dl + dprob["notes"]
        self._dec_head_backward(p, g, c["dec_notes"], dl, "dec.notes.out", np.asarray(c["starts"]["notes"], dt), dzh)
        if cfg["meta_instrument"]:
            It = np.asarray(b["I"], dt)
            dl = (_cce_grad_logits(out["instr"], It) * (cfg["w_instr"] * c["g_instr"])[..., None]).transpose(1, 0, 2)
            if "instr" in dprob:
                dl = dl + dprob["instr"]
            self._dec_head_backward(p, g, c["dec_instr"], dl, "dec.instr.out",
                                    np.asarray(c["starts"]["instr"], dt), dzh)
        if cfg["meta_velocity"]:
            Vt = np.asarray(b["Vel"], dt)
            pv = out["vel"]
            dl = (2.0 * (pv - Vt) * pv * (1.0 - pv) * (cfg["w_vel"] * c["g_vel"])[..., None]).transpose(1, 0, 2)
            self._dec_head_backward(p, g, c["dec_vel"], dl, "dec.vel.out",
                                    np.asarray(c["starts"]["vel"], dt).reshape(-1, 1), dzh)
        if cfg["meta_held"]:
            Dt = np.asarray(b["Held"], dt)
            dl = (_cce_grad_logits(out["held"], Dt) * (cfg["w_held"] * c["g_held"])[..., None]).transpose(1, 0, 2)
            self._dec_head_backward(p, g, c["dec_held"], dl, "dec.held.out", np.asarray(c["starts"]["held"], dt), dzh)
        if cfg["meta_next"]:
            Nt = np.asarray(b["Next"], dt)
            dl = (_cce_grad_logits(out["next"], Nt) * (cfg["w_next"] * c["g_next"])[..., None]).transpose(1, 0, 2)
            self._dec_head_backward(p, g, c["dec_next"], dl, "dec.next.out", np.asarray(c["starts"]["next"], dt), dzh)
        dz = dzh[:, :Z].copy()
        if cfg["style"]:
            Ct = np.asarray(b["C"], dt)
            dz[:, :cfg["C"]] += _cce_grad_logits(c["p_style"], Ct) * (cfg["w_style"] * c["g_style"])[:, None]
        if cfg["signature"]:
            off = cfg["C"] if cfg["style"] else 0
            St, sig = np.asarray(b["S"], dt), c["sig"]
            dz[:, off:off + cfg["SD"]] += (cfg["w_sig"] * c["g_sig"])[:, None] * 2.0 * (sig - St) / cfg["SD"] * (1.0 - sig ** 2)
        # latent
        mu, lv, eps = c["mu"], c["lv"], c["eps"]
        pvar = cfg["prior_std"] ** 2
        dmu = dz + cfg["beta"] * (mu - cfg["prior_mean"]) / pvar / B
        dlv = dz * eps * 0.5 * np.exp(lv / 2.0) + cfg["beta"] * (-0.5) * (1.0 - np.exp(lv) / pvar) / B
        g["enc.zmean.W"], g["enc.zmean.b"] = c["h1"].T @ dmu, dmu.sum(0)
        g["enc.zlogvar.W"], g["enc.zlogvar.b"] = c["h2"].T @ dlv, dlv.sum(0)
        d1, d2 = dmu @ p["enc.zmean.W"].T, dlv @ p["enc.zlogvar.W"].T
        dh = np.concatenate([d1, d2], axis=1) if cfg["split"] else d1 + d2
        if cfg["extra_layer"]:
            dpre = dh * (1.0 - c["extra"] ** 2)
            g["enc.extra.W"], g["enc.extra.b"] = c["extra_in"].T @ dpre, dpre.sum(0)
            dh = dpre @ p["enc.extra.W"].T
        if "enc.pack.W" in p:
            dpre = dh * (1.0 - c["pack"] ** 2)
            g["enc.pack.W"], g["enc.pack.b"] = c["cat"].T @ dpre, dpre.sum(0)
            dh = dpre @ p["enc.pack.W"].T
        k = 0
        d_notes = dh[:, k:k + H]
        k += H
        if cfg["meta_instrument"]:
            self._enc_rnn_backward(p, g, "enc.instr", c["enc_instr"], None, dh[:, k:k + H], False)
            k += H
        if cfg["meta_velocity"]:
            self._enc_rnn_backward(p, g, "enc.vel", c["enc_vel"], None, dh[:, k:k + H], False)
            k += H
        if cfg["meta_held"]:
            self._enc_rnn_backward(p, g, "enc.held", c["enc_held"], None, dh[:, k:k + H], False)
            k += H
        layers = enc_notes_layers(cfg)
        recs = {r[0]: r for r in c["enc_notes"]}
        dx, dlast = None, d_notes                 # dx: gradient w.r.t. the (time-ordered) input sequence of the layer above
        for li in range(len(layers) - 1, -1, -1):
            acc = None
            for j, (prefix, rev, _) in enumerate(layers[li]):
                dext = None
                if dx is not None:
                    dext = dx[..., j * H:(j + 1) * H] if len(layers[li]) > 1 else dx
                    dext = dext[::-1] if rev else dext
                _, _, xin, hs, cs, acts = recs[prefix]
                d_in = self._enc_rnn_backward(p, g, prefix, (xin, hs, cs, acts), dext, dlast, li > 0)
                if d_in is not None:
                    d_in = d_in[::-1] if rev else d_in
                    acc = d_in if acc is None else acc + d_in
            dx, dlast = acc, None
        return g

    # ---- optimizer ------------------------------------------------------------------------------
    def new_opt_state(self, p):
        return dict(t=0, m={k: np.zeros_like(v) for k, v in p.items()}, v={k: np.zeros_like(v) for k, v in p.items()})

    def opt_step(self, p, g, st, b1=0.9, b2=0.999, eps=1e-8, rho=0.9):
        """Keras 2.0.8 Adam (epsilon OUTSIDE the bias correction) or RMSprop; updates p in place (Appendix A.8)."""
        lr = self.cfg["lr"]
        st["t"] += 1
        t = st["t"]
        if self.cfg["optimizer"] == "Adam":
            lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
            for k in p:
                st["m"][k] = b1 * st["m"][k] + (1.0 - b1) * g[k]
                st["v"][k] = b2 * st["v"][k] + (1.0 - b2) * g[k] ** 2
                p[k] -= lr_t * st["m"][k] / (np.sqrt(st["v"][k]) + eps)
        elif self.cfg["optimizer"] == "RMSprop":
            for k in p:
                st["v"][k] = rho * st["v"][k] + (1.0 - rho) * g[k] ** 2
                p[k] -= lr * g[k] / (np.sqrt(st["v"][k]) + eps)
        else:
            raise ValueError(self.cfg["optimizer"])

    def train_step(self, p, st, batch, eps):
        m, c = self.forward(p, batch, eps)
        g = self.backward(p, c)
        self.opt_step(p, g, st)
        return m


def history_from_z(z):
    """History latent = previous window's z, zeros for the first window (reference vae_training.py:795-798)."""
    H = np.zeros_like(z)
    H[1:] = z[:-1]
    return H
