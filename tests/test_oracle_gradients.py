"""The oracle's analytic backward vs (i) central finite differences and (ii) an independent restatement of the
forward in torch float64 differentiated by autograd.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import vae_oracle as vo
from tests.oracle_util import tiny_problem


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "SimpleRNN"])
def test_finite_differences(cell):
    cfg, p, batch, eps, m = tiny_problem(cell, seed=3)
    met, c = m.forward(p, batch, eps)
    g = m.backward(p, c)
    assert set(g) == set(p)
    rng = np.random.default_rng(1)
    h = 1e-6
    for k in p:
        for _ in range(4):
            ix = tuple(int(rng.integers(0, s)) for s in p[k].shape)
            old = p[k][ix]
            p[k][ix] = old + h
            lp = m.forward(p, batch, eps)[0]["loss"]
            p[k][ix] = old - h
            lm = m.forward(p, batch, eps)[0]["loss"]
            p[k][ix] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - g[k][ix]) < 1e-7 + 1e-5 * abs(fd), (k, ix, fd, g[k][ix])


def _hs(x):
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def _rnn_t(cell, xp, U, h, c):
    H = U.shape[0]
    out = []
    for t in range(xp.shape[0]):
        if cell == "GRU":
            a = xp[t][:, :2 * H] + h @ U[:, :2 * H]
            z, r = _hs(a[:, :H]), _hs(a[:, H:])
            hh = torch.tanh(xp[t][:, 2 * H:] + (r * h) @ U[:, 2 * H:])
            h = z * h + (1 - z) * hh
        elif cell == "LSTM":
            a = xp[t] + h @ U
            i, f, g, o = _hs(a[:, :H]), _hs(a[:, H:2 * H]), torch.tanh(a[:, 2 * H:3 * H]), _hs(a[:, 3 * H:])
            c = f * c + i * g
            h = o * torch.tanh(c)
        else:
            h = torch.tanh(xp[t] + h @ U)
        out.append(h)
    return torch.stack(out)


def _torch_loss(cfg, P, b, eps):
    """Independent forward (torch ops, autograd-friendly); mirrors reference vae_definition.py graph."""
    cell, H, Z = cfg["cell"], cfg["H"], cfg["Z"]
    tt = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    X, I, Vel, Hist, Y, C = (tt(b[k]) for k in ("X", "I", "Vel", "Hist", "Y", "C"))
    B = X.shape[0]
    zero = torch.zeros(B, H, dtype=torch.float64)

    def enc(prefix, x):
        return _rnn_t(cell, x.transpose(0, 1) @ P[prefix + ".W"] + P[prefix + ".b"], P[prefix + ".U"], zero, zero)

    x = X
    for l in range(cfg["Le"]):
        seq = enc("enc.notes.%d" % l, x)
        x = seq.transpose(0, 1)
    feats = [seq[-1], enc("enc.instr", I)[-1], enc("enc.vel", Vel)[-1]]
    h = torch.tanh(torch.cat(feats, 1) @ P["enc.pack.W"] + P["enc.pack.b"])
    h = torch.tanh(h @ P["enc.extra.W"] + P["enc.extra.b"])
    mu = h[:, :H // 2] @ P["enc.zmean.W"] + P["enc.zmean.b"]
    lv = h[:, H // 2:] @ P["enc.zlogvar.W"] + P["enc.zlogvar.b"]
    z = mu + torch.exp(lv / 2) * tt(eps)
    kl = (cfg["beta"] * (-0.5 * torch.sum(1 + lv - mu ** 2 - torch.exp(lv), 1))).mean()
    zh = torch.cat([z, Hist], 1)

    def head(cells, inits, out, start, steps):
        xseq = None
        for l, (cp, ip) in enumerate(zip(cells, inits)):
            st = [torch.tanh(zh @ P["%s.%d.W" % (ip, s)] + P["%s.%d.b" % (ip, s)]) for s in range(vo.NSTATE[cell])]
            if l == 0:
                xp = (start @ P[cp + ".W"] + P[cp + ".b"]).unsqueeze(0).expand(steps, -1, -1)
            else:
                xp = xseq @ P[cp + ".W"] + P[cp + ".b"]
            xseq = _rnn_t(cell, xp, P[cp + ".U"], st[0], st[1] if cell == "LSTM" else None)
        return (xseq @ P[out + ".W"] + P[out + ".b"]).transpose(0, 1)

    Ld = cfg["Ld"]
    pn = torch.softmax(head(["dec.notes.%d" % l for l in range(Ld)], ["dec.notes.init.%d" % l for l in range(Ld)],
                            "dec.notes.out", torch.zeros(B, cfg["Dout"], dtype=torch.float64), cfg["T"]), -1)
    pi = torch.softmax(head(["dec.instr.cell"], ["dec.instr.init"], "dec.instr.out",
                            torch.zeros(B, cfg["ID"], dtype=torch.float64), cfg["V"]), -1)
    pv = torch.sigmoid(head(["dec.vel.cell"], ["dec.vel.init"], "dec.vel.out",
                            torch.zeros(B, 1, dtype=torch.float64), cfg["T"]))

    def cce(p, y):
        q = torch.clamp(p / p.sum(-1, keepdim=True), 1e-7, 1 - 1e-7)
        return -(y * torch.log(q)).sum(-1)

    w = tt(b["w_notes"])
    ln = (cce(pn, Y) * w).mean() / (w != 0).double().mean()
    li = cce(pi, I).mean(1).mean()
    lvel = ((pv - Vel) ** 2).mean(-1).mean(1).mean()
    ls = cce(torch.softmax(z[:, :cfg["C"]], -1), C).mean()
    return ln + cfg["w_instr"] * li + cfg["w_vel"] * lvel + cfg["w_style"] * ls + kl


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "SimpleRNN"])
def test_against_torch_autograd(cell):
    cfg, p, batch, eps, m = tiny_problem(cell, B=4, H=6, T=7, seed=5)
    met, c = m.forward(p, batch, eps)
    g = m.backward(p, c)
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    loss = _torch_loss(cfg, P, batch, eps)
    assert abs(loss.item() - met["loss"]) < 1e-12
    loss.backward()
    for k in p:
        gt = P[k].grad.numpy() if P[k].grad is not None else np.zeros_like(p[k])
        assert np.allclose(g[k], gt, rtol=1e-9, atol=1e-12), k


def test_training_reduces_loss():
    cfg, p, batch, eps, m = tiny_problem("GRU", seed=7, lr=1e-2)
    st = m.new_opt_state(p)
    first = m.train_step(p, st, batch, eps)["loss"]
    for _ in range(30):
        last = m.train_step(p, st, batch, eps)["loss"]
    assert last < first
