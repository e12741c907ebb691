"""torch-CPU port of the oracle's train step  --  TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT.

Only tests/ and bench.py's ``cpu_baseline`` leg may import this module (same rule as oracle/vae_oracle.py).

Why it exists: SURVEY section 8(d) asks for the CPU path timed beside the GPU number with ALL host cores at the largest batch
that finishes in under a minute per step.  The reference's own Keras CPU path cannot run here (Keras / TF-1 / recurrentshop are
absent), and the float64 NumPy oracle is written for clarity - its element-wise gate arithmetic is single-threaded.  This file
is the same algorithm (forward, analytic backward, Keras-Adam; SURVEY Appendix A, reference vae_definition.py:443-734) in
float32 torch tensor operations, which use every core for the GEMMs AND the element-wise work.  It covers the benched
configuration (all default heads on: instrument, velocity, style; pack / extra Dense; split; history) for GRU and LSTM cells.
``tests/test_torch_port_cpu.py`` pins it to the NumPy oracle (losses and every gradient), so the baseline number is the
oracle's arithmetic, not an approximation of it.
"""
from __future__ import annotations

import torch

CE_EPS = 1e-7


def _hs(x):
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def _dhs(y):
    return 0.2 * ((y > 0.0) & (y < 1.0)).to(y.dtype)


def rnn_forward(cell, xp, U, h0, c0=None, const=False):
    """xp (T,B,G*H) (or (B,G*H) with ``const``: the same row every step, Appendix A.6); returns hs (T+1,B,H), cs, acts"""
    T = xp.shape[0] if not const else const
    B, H = h0.shape
    GH = U.shape[1]
    hs = torch.empty((T + 1, B, H), dtype=h0.dtype)
    hs[0] = h0
    acts = torch.empty((T, B, GH), dtype=h0.dtype)
    cs = None
    if cell == "LSTM":
        cs = torch.empty((T + 1, B, H), dtype=h0.dtype)
        cs[0] = c0
    U_zr, U_h = (U[:, :2 * H].contiguous(), U[:, 2 * H:].contiguous()) if cell == "GRU" else (None, None)
    for t in range(T):
        x = xp if const else xp[t]
        h = hs[t]
        if cell == "GRU":
            a = torch.addmm(x[:, :2 * H], h, U_zr)
            zr = _hs(a)
            z, r = zr[:, :H], zr[:, H:]
            hh = torch.tanh(torch.addmm(x[:, 2 * H:], r * h, U_h))
            hs[t + 1] = z * h + (1.0 - z) * hh
            acts[t, :, :2 * H] = zr
            acts[t, :, 2 * H:] = hh
        else:
            a = torch.addmm(x, h, U)
            i, f, o = _hs(a[:, :H]), _hs(a[:, H:2 * H]), _hs(a[:, 3 * H:])
            g = torch.tanh(a[:, 2 * H:3 * H])
            cs[t + 1] = f * cs[t] + i * g
            hs[t + 1] = o * torch.tanh(cs[t + 1])
            acts[t, :, :H], acts[t, :, H:2 * H], acts[t, :, 2 * H:3 * H], acts[t, :, 3 * H:] = i, f, g, o
    return hs, cs, acts


def rnn_backward(cell, hs, cs, acts, U, dhs_ext=None, dh_last=None):
    """BPTT (oracle/vae_oracle.py:rnn_backward).  Returns da (T,B,G*H), dh0, dc0; dU is formed by the caller as ONE GEMM over
    all steps (hs[:-1]^T da), which is how any serious CPU implementation would batch it."""
    T, B, GH = acts.shape
    H = U.shape[0]
    da = torch.empty_like(acts)
    dh = torch.zeros((B, H), dtype=acts.dtype) if dh_last is None else dh_last.clone()
    dc = torch.zeros((B, H), dtype=acts.dtype)
    Ut = U.t().contiguous()
    Ut_zr, Ut_h = (Ut[:2 * H].contiguous(), Ut[2 * H:].contiguous()) if cell == "GRU" else (None, None)
    rh = torch.empty((T, B, H), dtype=acts.dtype) if cell == "GRU" else None
    for t in range(T - 1, -1, -1):
        d = dh + dhs_ext[t] if dhs_ext is not None else dh
        hp = hs[t]
        if cell == "GRU":
            z, r, hh = acts[t, :, :H], acts[t, :, H:2 * H], acts[t, :, 2 * H:]
            da_h = d * (1.0 - z) * (1.0 - hh * hh)
            drh = da_h @ Ut_h
            da[t, :, :H] = d * (hp - hh) * _dhs(z)
            da[t, :, H:2 * H] = drh * hp * _dhs(r)
            da[t, :, 2 * H:] = da_h
            rh[t] = r * hp
            dh = d * z + drh * r + da[t, :, :2 * H] @ Ut_zr
        else:
            i, f, g, o = acts[t, :, :H], acts[t, :, H:2 * H], acts[t, :, 2 * H:3 * H], acts[t, :, 3 * H:]
            tc = torch.tanh(cs[t + 1])
            dct = dc + d * o * (1.0 - tc * tc)
            da[t, :, :H] = dct * g * _dhs(i)
            da[t, :, H:2 * H] = dct * cs[t] * _dhs(f)
            da[t, :, 2 * H:3 * H] = dct * i * (1.0 - g * g)
            da[t, :, 3 * H:] = d * tc * _dhs(o)
            dc = dct * f
            dh = da[t] @ Ut
    return da, dh, dc, rh


def _dU(cell, hs, da, rh):
    T, B, GH = da.shape
    H = hs.shape[2]
    hp = hs[:-1].reshape(T * B, H)
    d2 = da.reshape(T * B, GH)
    if cell == "GRU":
        return torch.cat([hp.t() @ d2[:, :2 * H], rh.reshape(T * B, H).t() @ d2[:, 2 * H:]], 1)
    return hp.t() @ d2


def _cce(p, y):
    q = torch.clamp(p / p.sum(-1, keepdim=True), CE_EPS, 1.0 - CE_EPS)
    return -(y * torch.log(q)).sum(-1)


def _cce_grad_logits(p, y):
    inside = ((p >= CE_EPS) & (p <= 1.0 - CE_EPS)).to(p.dtype)
    ym = y * inside
    return p * ym.sum(-1, keepdim=True) - ym


class TorchCPUVAE(object):
    """The benched graph (all default heads) on torch CPU tensors; parameters are a dict name -> tensor with the oracle's names."""

    def __init__(self, cfg, dtype=torch.float32):
        self.cfg, self.dtype = dict(cfg), dtype
        need = dict(meta_instrument=True, meta_velocity=True, extra_layer=True, split=True, history=True, style=True)
        for k, v in need.items():
            if cfg[k] != v:
                raise NotImplementedError("torch_cpu covers the benched configuration only (%s=%r)" % (k, cfg[k]))
        if cfg["cell"] not in ("GRU", "LSTM"):
            raise NotImplementedError(cfg["cell"])

    def tensors(self, named):
        return {k: torch.as_tensor(v, dtype=self.dtype).clone() for k, v in named.items()}

    # ---- forward + losses ----------------------------------------------------------------------------------------
    def forward(self, p, b, eps):
        cfg, cell, H, Z, T, V = self.cfg, self.cfg["cell"], self.cfg["H"], self.cfg["Z"], self.cfg["T"], self.cfg["V"]
        lstm = cell == "LSTM"
        t_ = lambda a: torch.as_tensor(a, dtype=self.dtype)
        X, I, Vel, Hist, Y, Cc = (t_(b[k]) for k in ("X", "I", "Vel", "Hist", "Y", "C"))
        B = X.shape[0]
        c = dict(B=B, X=X, I=I, Vel=Vel, Y=Y, C=Cc)
        zero = torch.zeros((B, H), dtype=self.dtype)

        def enc(prefix, x_tm):
            xp = x_tm @ p[prefix + ".W"] + p[prefix + ".b"]
            return (x_tm,) + rnn_forward(cell, xp, p[prefix + ".U"], zero, zero if lstm else None)

        x = X.transpose(0, 1).contiguous()
        c["enc_notes"] = []
        for l in range(cfg["Le"]):
            rec = enc("enc.notes.%d" % l, x)
            c["enc_notes"].append(rec)
            x = rec[1][1:]
        c["enc_instr"] = enc("enc.instr", I.transpose(0, 1).contiguous())
        c["enc_vel"] = enc("enc.vel", Vel.transpose(0, 1).contiguous())
        cat = torch.cat([c["enc_notes"][-1][1][-1], c["enc_instr"][1][-1], c["enc_vel"][1][-1]], 1)
        pack = torch.tanh(cat @ p["enc.pack.W"] + p["enc.pack.b"])
        extra = torch.tanh(pack @ p["enc.extra.W"] + p["enc.extra.b"])
        h1, h2 = extra[:, :H // 2], extra[:, H // 2:]
        mu = h1 @ p["enc.zmean.W"] + p["enc.zmean.b"]
        lv = h2 @ p["enc.zlogvar.W"] + p["enc.zlogvar.b"]
        eps = t_(eps)
        z = mu + torch.exp(lv / 2.0) * eps
        zh = torch.cat([z, Hist], 1)
        c.update(cat=cat, pack=pack, extra=extra, h1=h1, h2=h2, mu=mu, lv=lv, eps=eps, z=z, zh=zh)
        ns = 2 if lstm else 1

        def head(key, cells, inits, out, start, steps):
            layers, xseq = [], None
            for l, (cp, ip) in enumerate(zip(cells, inits)):
                st = [torch.tanh(zh @ p["%s.%d.W" % (ip, s)] + p["%s.%d.b" % (ip, s)]) for s in range(ns)]
                if l == 0:
                    xp0 = start @ p[cp + ".W"] + p[cp + ".b"]
                    hs, cs, acts = rnn_forward(cell, xp0, p[cp + ".U"], st[0], st[1] if lstm else None, const=steps)
                else:
                    hs, cs, acts = rnn_forward(cell, xseq @ p[cp + ".W"] + p[cp + ".b"], p[cp + ".U"], st[0], st[1] if lstm else None)
                layers.append((cp, ip, xseq, hs, cs, acts))
                xseq = hs[1:]
            c[key] = layers
            return xseq @ p[out + ".W"] + p[out + ".b"]

        Ld = cfg["Ld"]
        s_n, s_i, s_v = (torch.zeros((B, cfg["Dout"]), dtype=self.dtype), torch.zeros((B, cfg["ID"]), dtype=self.dtype),
                         torch.zeros((B, 1), dtype=self.dtype))
        c["starts"] = (s_n, s_i, s_v)
        pn = torch.softmax(head("dec_notes", ["dec.notes.%d" % l for l in range(Ld)], ["dec.notes.init.%d" % l for l in range(Ld)],
                                "dec.notes.out", s_n, T), -1)                 # (T,B,D) time-major
        pi = torch.softmax(head("dec_instr", ["dec.instr.cell"], ["dec.instr.init"], "dec.instr.out", s_i, V), -1)
        pv = torch.sigmoid(head("dec_vel", ["dec.vel.cell"], ["dec.vel.init"], "dec.vel.out", s_v, T))
        c.update(pn=pn, pi=pi, pv=pv)
        Yt, It, Vt = Y.transpose(0, 1), I.transpose(0, 1), Vel.transpose(0, 1)
        m = {}
        m["kl"] = (cfg["beta"] * (-0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv), 1))).mean()     # prior N(0,1)
        m["notes_loss"] = _cce(pn, Yt).mean()
        m["instr_loss"] = _cce(pi, It).mean()
        m["vel_loss"] = ((pv - Vt) ** 2).mean()
        ps = torch.softmax(z[:, :cfg["C"]], -1)
        c["ps"] = ps
        m["style_loss"] = _cce(ps, Cc).mean()
        m["loss"] = (m["notes_loss"] + m["kl"] + cfg["w_instr"] * m["instr_loss"] + cfg["w_vel"] * m["vel_loss"] +
                     cfg["w_style"] * m["style_loss"])
        return {k: float(v) for k, v in m.items()}, c

    # ---- backward ----------------------------------------------------------------------------------------------
    def backward(self, p, c):
        cfg, cell, H, Z, T, V = self.cfg, self.cfg["cell"], self.cfg["H"], self.cfg["Z"], self.cfg["T"], self.cfg["V"]
        lstm = cell == "LSTM"
        B = c["B"]
        g = {}
        zh = c["zh"]
        dzh = torch.zeros_like(zh)
        Yt, It, Vt = c["Y"].transpose(0, 1), c["I"].transpose(0, 1), c["Vel"].transpose(0, 1)

        def head_bwd(layers, dl, out, start):
            top = layers[-1][3][1:]
            R = top.shape[0] * B
            g[out + ".W"] = top.reshape(R, H).t() @ dl.reshape(R, -1)
            g[out + ".b"] = dl.sum((0, 1))
            dext = dl @ p[out + ".W"].t()
            for l in range(len(layers) - 1, -1, -1):
                cp, ip, xseq, hs, cs, acts = layers[l]
                da, dh0, dc0, rh = rnn_backward(cell, hs, cs, acts, p[cp + ".U"], dext)
                g[cp + ".U"] = _dU(cell, hs, da, rh)
                g[cp + ".b"] = da.sum((0, 1))
                if l == 0:
                    g[cp + ".W"] = start.t() @ da.sum(0)
                else:
                    g[cp + ".W"] = xseq.reshape(R, H).t() @ da.reshape(R, -1)
                    dext = da @ p[cp + ".W"].t()
                for s, dst in enumerate([dh0, dc0][:2 if lstm else 1]):
                    st = (hs if s == 0 else cs)[0]
                    dpre = dst * (1.0 - st * st)
                    g["%s.%d.W" % (ip, s)] = zh.t() @ dpre
                    g["%s.%d.b" % (ip, s)] = dpre.sum(0)
                    dzh.add_(dpre @ p["%s.%d.W" % (ip, s)].t())

        s_n, s_i, s_v = c["starts"]
        head_bwd(c["dec_notes"], _cce_grad_logits(c["pn"], Yt) / (B * T), "dec.notes.out", s_n)
        head_bwd(c["dec_instr"], _cce_grad_logits(c["pi"], It) * (cfg["w_instr"] / (B * V)), "dec.instr.out", s_i)
        pv = c["pv"]
        head_bwd(c["dec_vel"], 2.0 * (pv - Vt) * pv * (1.0 - pv) * (cfg["w_vel"] / (B * T)), "dec.vel.out", s_v)
        dz = dzh[:, :Z].clone()
        dz[:, :cfg["C"]] += _cce_grad_logits(c["ps"], c["C"]) * (cfg["w_style"] / B)
        mu, lv, eps = c["mu"], c["lv"], c["eps"]
        dmu = dz + cfg["beta"] * mu / B
        dlv = dz * eps * 0.5 * torch.exp(lv / 2.0) + cfg["beta"] * (-0.5) * (1.0 - torch.exp(lv)) / B
        g["enc.zmean.W"], g["enc.zmean.b"] = c["h1"].t() @ dmu, dmu.sum(0)
        g["enc.zlogvar.W"], g["enc.zlogvar.b"] = c["h2"].t() @ dlv, dlv.sum(0)
        dh = torch.cat([dmu @ p["enc.zmean.W"].t(), dlv @ p["enc.zlogvar.W"].t()], 1)
        dpre = dh * (1.0 - c["extra"] ** 2)
        g["enc.extra.W"], g["enc.extra.b"] = c["pack"].t() @ dpre, dpre.sum(0)
        dh = dpre @ p["enc.extra.W"].t()
        dpre = dh * (1.0 - c["pack"] ** 2)
        g["enc.pack.W"], g["enc.pack.b"] = c["cat"].t() @ dpre, dpre.sum(0)
        dh = dpre @ p["enc.pack.W"].t()

        def enc_bwd(prefix, rec, dext, dlast, need_dx):
            x, hs, cs, acts = rec
            da, _, _, rh = rnn_backward(cell, hs, cs, acts, p[prefix + ".U"], dext, dlast)
            R = da.shape[0] * B
            g[prefix + ".U"] = _dU(cell, hs, da, rh)
            g[prefix + ".b"] = da.sum((0, 1))
            g[prefix + ".W"] = x.reshape(R, -1).t() @ da.reshape(R, -1)
            return da @ p[prefix + ".W"].t() if need_dx else None

        enc_bwd("enc.instr", c["enc_instr"], None, dh[:, H:2 * H], False)
        enc_bwd("enc.vel", c["enc_vel"], None, dh[:, 2 * H:3 * H], False)
        dext, dlast = None, dh[:, :H]
        for l in range(cfg["Le"] - 1, -1, -1):
            dext = enc_bwd("enc.notes.%d" % l, c["enc_notes"][l], dext, dlast, l > 0)
            dlast = None
        return g

    # ---- Keras Adam (Appendix A.8) ---------------------------------------------------------------------------
    def new_opt_state(self, p):
        return dict(t=0, m={k: torch.zeros_like(v) for k, v in p.items()}, v={k: torch.zeros_like(v) for k, v in p.items()})

    def opt_step(self, p, g, st, b1=0.9, b2=0.999, eps=1e-8):
        st["t"] += 1
        t = st["t"]
        lr_t = self.cfg["lr"] * (1.0 - b2 ** t) ** 0.5 / (1.0 - b1 ** t)
        for k in p:
            st["m"][k].mul_(b1).add_(g[k], alpha=1.0 - b1)
            st["v"][k].mul_(b2).addcmul_(g[k], g[k], value=1.0 - b2)
            p[k].sub_(lr_t * st["m"][k] / (st["v"][k].sqrt() + eps))

    def train_step(self, p, st, batch, eps):
        with torch.no_grad():
            m, c = self.forward(p, batch, eps)
            g = self.backward(p, c)
            self.opt_step(p, g, st)
        return m


def timed_sample(cell, T, B, V, Z, C, budget_s=15.0, threads=0):
    """windows/s of the full train step at B windows, from a BOUNDED sample: the same step on the first T_s of the T time steps
    (every cost of the step but the latent block and the optimizer is linear in T, so the rate is scaled by T_s / T).  T_s is the
    largest power of two whose two steps (one untimed, one timed) are expected to fit ``budget_s`` after a T_s = 8 calibration.
    Returns a dict for bench.py's ``cpu_baseline``."""
    import os
    import time

    import numpy as np

    from midi_vae_amd.layout import ModelSpec, init_params
    from midi_vae_amd.synth import make_windows
    from oracle.vae_oracle import make_cfg
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    oh = lambda idx, n: np.eye(n, dtype=np.float32)[idx.astype(np.int64)]

    def run(Ts, reps):
        spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=Ts, V=V, ID=16, C=C, Le=2, Ld=2)
        tv = TorchCPUVAE(make_cfg(**spec.oracle_cfg()))
        w = make_windows(B, Ts, 61, V, 16, C, Z, seed=1234, epsilon_std=spec.epsilon_std)
        batch = dict(X=oh(w["x_idx"], 61), I=oh(w["i_idx"], 16), Vel=w["vel"][..., None], Hist=w["hist"], Y=oh(w["x_idx"], 61),
                     C=oh(w["c_idx"], C))
        P = tv.tensors(init_params(spec, 1234))
        st = tv.new_opt_state(P)
        tv.train_step(P, st, batch, w["eps"])                 # untimed: thread pools, page faults
        t0 = time.perf_counter()
        for _ in range(reps):
            tv.train_step(P, st, batch, w["eps"])
        return (time.perf_counter() - t0) / reps

    # thread count: all cores is not always the fastest for 10^5 small tensor operations per step - calibrate
    best = None
    for n in sorted({cores, min(cores, 64), min(cores, 16)} if not threads else {threads}, reverse=True):
        torch.set_num_threads(n)
        t8 = run(8, 1)
        if best is None or t8 < best[1]:
            best = (n, t8)
    n_thr, t8 = best
    torch.set_num_threads(n_thr)
    Ts = 8
    while Ts * 2 <= T and t8 * (Ts * 2 / 8.0) * 2 <= budget_s:
        Ts *= 2
    dt = run(Ts, 1) if Ts > 8 else t8
    step_s = dt * T / Ts
    return {"value": B / step_s, "unit": "windows/s", "cores": n_thr, "host_cores": cores, "kind": "port",
            "sample": "the identical train step (%s, H=256, B=%d windows) in float32 torch-CPU tensor operations on %d threads "
                      "(oracle/torch_cpu.py), timed on the first %d of the T=%d time steps: %.2f s, scaled by T/T_s to %.1f s per "
                      "full step (thread count picked from {all, 64, 16} by a T_s=8 calibration: %.2f s)"
                      % (cell, B, n_thr, Ts, T, dt, step_s, t8)}


def elbo_inputs(*a, **k):
    """data generator shared with bench.py's GPU leg (midi_vae_amd.synth.elbo_inputs: no arithmetic)"""
    from midi_vae_amd.synth import elbo_inputs as f
    return f(*a, **k)


def elbo_epsilon(*a, **k):
    from midi_vae_amd.synth import elbo_epsilon as f
    return f(*a, **k)


def elbo_trajectory(cell, T, B, V, Z, C, steps, threads=16):
    """``steps`` real optimizer steps (forward, analytic backward, Keras-Adam) in float64 torch-CPU arithmetic on the first B windows
    of the bench's inputs, a fresh epsilon per step: the ELBO (Keras total loss) and its parts after every step - the CPU side of
    bench.py's ``elbo`` block (north_star: 'ELBO within 1e-3 of reference after equal steps')."""
    import numpy as np

    from oracle.vae_oracle import make_cfg
    torch.set_num_threads(threads)
    spec, w, params = elbo_inputs(cell, T, B, V, Z, C)
    oh = lambda idx, n: np.eye(n)[idx.astype(np.int64)]
    tv = TorchCPUVAE(make_cfg(**spec.oracle_cfg()), dtype=torch.float64)
    batch = dict(X=oh(w["x_idx"], 61), I=oh(w["i_idx"], 16), Vel=w["vel"][..., None].astype(np.float64), Hist=w["hist"].astype(np.float64),
                 Y=oh(w["x_idx"], 61), C=oh(w["c_idx"], C))
    P = tv.tensors(params)
    st = tv.new_opt_state(P)
    out = []
    for i in range(steps):
        m = tv.train_step(P, st, batch, elbo_epsilon(i, B, Z, spec.epsilon_std).astype(np.float64))
        out.append({k: float(m[k]) for k in ("loss", "notes_loss", "instr_loss", "vel_loss", "style_loss", "kl")})
    return out


if __name__ == "__main__":
    import argparse
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import midi_vae_amd  # noqa: F401
    ap = argparse.ArgumentParser()
    ap.add_argument("--cell", default="LSTM")
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--V", type=int, default=4)
    ap.add_argument("--Z", type=int, default=64)
    ap.add_argument("--C", type=int, default=2)
    ap.add_argument("--budget", type=float, default=15.0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--elbo-steps", type=int, default=0, help="> 0: print the ELBO trajectory of that many optimizer steps instead")
    a = ap.parse_args()
    if a.elbo_steps:
        print(json.dumps(elbo_trajectory(a.cell, a.T, a.B, a.V, a.Z, a.C, a.elbo_steps, a.threads or 16)))
    else:
        print(json.dumps(timed_sample(a.cell, a.T, a.B, a.V, a.Z, a.C, a.budget, a.threads)))
