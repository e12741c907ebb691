"""oracle/torch_cpu.py (the all-cores CPU baseline bench.py times) against the NumPy oracle: losses, every gradient, and the
parameters after two Keras-Adam steps.  CPU only."""
import numpy as np
import pytest
import torch

from oracle.torch_cpu import TorchCPUVAE
from tests.oracle_util import tiny_problem


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_torch_port_matches_numpy_oracle(cell):
    cfg, p, batch, eps, m = tiny_problem(cell, B=5, H=8, Z=6, T=7, V=3, seed=4)
    batch.pop("w_notes")                                  # the benched step uses unit sample weights
    met, c = m.forward(p, batch, eps)
    g = m.backward(p, c)
    tv = TorchCPUVAE(cfg, dtype=torch.float64)
    P = tv.tensors(p)
    with torch.no_grad():
        mt, ct = tv.forward(P, batch, eps)
        gt = tv.backward(P, ct)
    for k in mt:
        assert abs(mt[k] - met[k]) < 1e-12, (k, mt[k], met[k])
    assert set(gt) == set(g)
    for k in g:
        assert np.allclose(gt[k].numpy(), g[k], rtol=1e-9, atol=1e-12), k
    st_o, st_t = m.new_opt_state(p), tv.new_opt_state(P)
    for _ in range(2):
        m.train_step(p, st_o, batch, eps)
        tv.train_step(P, st_t, batch, eps)
    for k in p:
        assert np.allclose(P[k].numpy(), p[k], rtol=1e-9, atol=1e-12), k
