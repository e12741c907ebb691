"""Closed-form known answers for the oracle's primitives.  CPU only."""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from tests.oracle_util import tiny_problem


def test_hard_sigmoid_knots():
    x = np.array([-3.0, -2.5, 0.0, 2.5, 3.0, 1.0])
    assert np.allclose(vo.hard_sigmoid(x), [0, 0, 0.5, 1, 1, 0.7])
    assert np.allclose(vo._dhs(vo.hard_sigmoid(x)), [0, 0, 0.2, 0, 0, 0.2])


def test_kl_zero_at_prior_and_matches_closed_form():
    cfg, p, batch, eps, m = tiny_problem("GRU")
    for k in ("enc.zmean.W", "enc.zmean.b", "enc.zlogvar.W", "enc.zlogvar.b"):
        p[k][...] = 0.0
    met, c = m.forward(p, batch, eps * 0)
    assert abs(met["kl"]) < 1e-15                     # KL(N(0,1) || N(0,1)) = 0
    import torch
    mu = torch.tensor([[0.3, -1.2, 0.5]], dtype=torch.float64)
    lv = torch.tensor([[0.1, -0.7, 0.4]], dtype=torch.float64)
    kl_t = torch.distributions.kl_divergence(torch.distributions.Normal(mu, (lv / 2).exp()),
                                             torch.distributions.Normal(0.5, 2.0)).sum(1)
    pm, ps = 0.5, 2.0
    kl = -0.5 * np.sum(1 + lv.numpy() - 2 * np.log(ps) - ((mu.numpy() - pm) ** 2 + np.exp(lv.numpy())) / ps ** 2, 1)
    assert np.allclose(kl, kl_t.numpy())


def test_uniform_softmax_ce_is_log_k():
    p = np.full((2, 3, 61), 1.0 / 61)
    y = np.zeros_like(p)
    y[..., 7] = 1
    assert np.allclose(vo._cce(p, y), np.log(61.0))


def test_gru_zero_weights_halves_state():
    H, B, T = 4, 2, 3
    xp = np.zeros((T, B, 3 * H))
    h0 = np.arange(B * H, dtype=float).reshape(B, H)
    hs, _, acts = vo.rnn_forward("GRU", xp, np.zeros((H, 3 * H)), h0)
    for t in range(T):
        assert np.allclose(hs[t + 1], h0 * 0.5 ** (t + 1))   # z=0.5, hh=tanh(0)=0
    assert np.allclose(acts[..., :2 * H], 0.5)


def test_lstm_zero_weights():
    H, B = 3, 2
    xp = np.zeros((2, B, 4 * H))
    hs, cs, _ = vo.rnn_forward("LSTM", xp, np.zeros((H, 4 * H)), np.ones((B, H)), np.ones((B, H)))
    assert np.allclose(cs[1], 0.5) and np.allclose(hs[1], 0.5 * np.tanh(0.5))


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "SimpleRNN"])
def test_decoder_with_zero_start_ignores_input_kernel(cell):
    """F9 / Appendix A.6: the decoder's layer-0 input kernel multiplies the constant zero start vector."""
    cfg, p, batch, eps, m = tiny_problem(cell)
    met0, c = m.forward(p, batch, eps)
    g = m.backward(p, c)
    assert np.all(g["dec.notes.0.W"] == 0) and np.all(g["dec.instr.cell.W"] == 0) and np.all(g["dec.vel.cell.W"] == 0)
    p["dec.notes.0.W"] += 1.0
    met1, _ = m.forward(p, batch, eps)
    assert met0["loss"] == met1["loss"]


def test_weighted_mean_semantics():
    sc = np.array([[1.0, 2.0], [3.0, 4.0]])
    w = np.array([[1.0, 0.0], [0.5, 1.0]])
    loss, g = vo._weighted_mean(sc, w)
    assert np.isclose(loss, np.mean(sc * w) / 0.75)
    assert np.allclose(g, w / (0.75 * 4))


def test_keras_adam_first_step_is_lr_sign():
    cfg, p, batch, eps, m = tiny_problem("GRU")
    st = m.new_opt_state(p)
    g = {k: np.ones_like(v) * 3.0 for k, v in p.items()}
    p0 = {k: v.copy() for k, v in p.items()}
    m.opt_step(p, g, st)
    # t=1: lr_t = lr*sqrt(1-b2)/(1-b1); m=(1-b1)g; v=(1-b2)g^2 -> step = lr*|g|/(|g|+eps/sqrt(1-b2)) ~ lr
    lr_t = cfg["lr"] * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = lr_t * (0.1 * 3.0) / (np.sqrt(0.001 * 9.0) + 1e-8)
    for k in p:
        assert np.allclose(p0[k] - p[k], want)


def test_history_roll():
    z = np.arange(12.0).reshape(4, 3)
    h = vo.history_from_z(z)
    assert np.all(h[0] == 0) and np.array_equal(h[1:], z[:-1])


def test_readout_add_is_a_forward_only_study_switch_of_the_oracle():
    """SURVEY A.6 hedge (VERDICT r05 missing #3): cfg['readout'] = 'add' steps the decoder on x_t = start + y_{t-1}; with every
    output Dense zeroed except its bias the outputs are constant, and with zero output biases of a sigmoid / softmax head y is NOT
    zero - so the two readings differ even there; 'none' (the default, the reference as written) is what everything else tests.
    No backward pass exists for 'add'."""
    from oracle.vae_oracle import OracleVAE, make_cfg
    from tests.oracle_util import tiny_problem
    cfg, p, batch, eps, m = tiny_problem("GRU", B=3, H=8, Z=6, T=5, V=3, seed=2)
    assert cfg.get("readout", "none") == "none"
    base, cache = m.forward(p, batch, eps)
    m2 = OracleVAE(make_cfg(**dict(cfg, readout="add")))
    alt, cache2 = m2.forward(p, batch, eps)
    assert abs(alt["loss"] - base["loss"]) > 1e-6
    # step 0 of a head sees x_0 = start + start = 0 on both readings (the packers' start rows are zeros): the first output agrees
    np.testing.assert_allclose(cache2["out"]["notes"][:, 0], cache["out"]["notes"][:, 0], rtol=1e-12, atol=1e-14)
    assert np.abs(cache2["out"]["notes"][:, 1:] - cache["out"]["notes"][:, 1:]).max() > 1e-6
    with pytest.raises(NotImplementedError, match="forward-only"):
        m2.backward(p, cache2)
