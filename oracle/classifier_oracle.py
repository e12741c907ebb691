"""CPU oracle of the three style classifiers  --  TEST INFRASTRUCTURE, NOT PRODUCT (same rule as oracle/vae_oracle.py).

reference pitch_classifier.py:89-103, velocity_classifier.py:110-125, instrument_classifier.py:93-107: ``num_layers`` Keras
GRU(lstm_size) layers (all but the last return sequences; Keras-2.0.x GRU: tanh / hard_sigmoid, reset gate before the candidate
matmul - SURVEY Appendix A.2) -> Dense(num_classes, softmax); loss categorical_crossentropy, metric accuracy, Keras Adam.
**Parity unpinned** like the VAE oracle (Keras is not vendored): restated from the published layer semantics, float64 NumPy.
"""
from collections import OrderedDict

import numpy as np

from .vae_oracle import GATES, _cce, _cce_grad_logits, rnn_backward, rnn_forward, softmax


def param_shapes(cfg):
    G, H = GATES[cfg["cell"]], cfg["H"]
    P = OrderedDict()
    for l in range(cfg["L"]):
        P["rnn.%d.W" % l], P["rnn.%d.U" % l], P["rnn.%d.b" % l] = (cfg["K"] if l == 0 else H, G * H), (H, G * H), (G * H,)
    P["cls.out.W"], P["cls.out.b"] = (H, cfg["C"]), (cfg["C"],)
    return P


class OracleClassifier(object):
    def __init__(self, cfg):
        self.cfg = dict(cfg)

    def forward(self, p, X, Y=None):
        """X (B,T,K); Y (B,C) one-hot.  Returns (probs (B,C), metrics, cache)."""
        cfg = self.cfg
        x = np.asarray(X, np.float64).transpose(1, 0, 2)
        B = x.shape[1]
        z0 = np.zeros((B, cfg["H"]))
        recs = []
        for l in range(cfg["L"]):
            xp = x @ p["rnn.%d.W" % l] + p["rnn.%d.b" % l]
            hs, cs, acts = rnn_forward(cfg["cell"], xp, p["rnn.%d.U" % l], z0, z0 if cfg["cell"] == "LSTM" else None)
            recs.append((x, hs, cs, acts))
            x = hs[1:]
        h = recs[-1][1][-1]
        probs = softmax(h @ p["cls.out.W"] + p["cls.out.b"])
        m = {}
        if Y is not None:
            Y = np.asarray(Y, np.float64)
            m["loss"] = float(np.mean(_cce(probs, Y)))
            m["acc"] = float(np.mean(np.argmax(probs, -1) == np.argmax(Y, -1)))
        return probs, m, dict(recs=recs, h=h, probs=probs, Y=Y)

    def backward(self, p, c):
        cfg = self.cfg
        B = c["h"].shape[0]
        g = {}
        dl = _cce_grad_logits(c["probs"], c["Y"]) / B
        g["cls.out.W"], g["cls.out.b"] = c["h"].T @ dl, dl.sum(0)
        dlast, dext = dl @ p["cls.out.W"].T, None
        for l in range(cfg["L"] - 1, -1, -1):
            x, hs, cs, acts = c["recs"][l]
            da, dU, _, _ = rnn_backward(cfg["cell"], hs, cs, acts, p["rnn.%d.U" % l], dext, dlast)
            g["rnn.%d.U" % l], g["rnn.%d.b" % l] = dU, da.sum((0, 1))
            g["rnn.%d.W" % l] = np.einsum("tbk,tbn->kn", x, da)
            dext, dlast = (da @ p["rnn.%d.W" % l].T if l > 0 else None), None
        return g

    # ---- Keras 2.0.8 Adam (epsilon outside the bias correction) --------------------------------
    def new_opt_state(self, p):
        return dict(t=0, m={k: np.zeros_like(v) for k, v in p.items()}, v={k: np.zeros_like(v) for k, v in p.items()})

    def opt_step(self, p, g, st, lr, b1=0.9, b2=0.999, eps=1e-8):
        st["t"] += 1
        t = st["t"]
        lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        for k in p:
            st["m"][k] = b1 * st["m"][k] + (1.0 - b1) * g[k]
            st["v"][k] = b2 * st["v"][k] + (1.0 - b2) * g[k] ** 2
            p[k] -= lr_t * st["m"][k] / (np.sqrt(st["v"][k]) + eps)
