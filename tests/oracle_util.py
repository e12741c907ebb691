"""Shared helpers for oracle-based tests (test infrastructure)."""
import numpy as np

from oracle.vae_oracle import OracleVAE, make_cfg, param_shapes


def onehot(idx, n):
    out = np.zeros(idx.shape + (n,))
    np.put_along_axis(out, idx[..., None], 1, -1)
    return out


def tiny_problem(cell="GRU", B=3, H=8, Z=6, T=5, V=2, ID=3, D=7, C=2, seed=0, scale=0.3, **kw):
    cfg = make_cfg(cell=cell, H=H, Z=Z, T=T, V=V, ID=ID, Din=D, Dout=D, C=C, **kw)
    rng = np.random.default_rng(seed)
    p = {k: rng.standard_normal(s) * scale for k, s in param_shapes(cfg).items()}
    X = onehot(rng.integers(0, D, (B, T)), D)
    batch = dict(X=X, I=onehot(rng.integers(0, ID, (B, V)), ID), Vel=rng.random((B, T, 1)),
                 Hist=rng.standard_normal((B, Z)) * 0.1, Y=X, C=onehot(rng.integers(0, C, (B,)), C),
                 w_notes=np.where(rng.random((B, T)) < 0.3, 0.5, 1.0))
    eps = rng.standard_normal((B, Z)) * 0.01
    return cfg, p, batch, eps, OracleVAE(cfg)
