"""The oracle's cell skeletons and its optimizer against an implementation this repository did not write: torch.nn.LSTMCell /
RNNCell / GRUCell and torch.optim.Adam (VERDICT r05 'weak 1': everything else the oracle is pinned by was written here).

Keras (~2.0.8) and recurrentshop are not in /root/reference and cannot be installed, so the reference's own arithmetic stays
unpinned; what CAN be anchored is that oracle/vae_oracle.py's recurrences are the textbook cells with the documented differences
and nothing else:
  * LSTM  = torch.nn.LSTMCell once the recurrent activation is sigmoid instead of Keras' hard_sigmoid (test-only switch
            ``rec_act``), gate order [i | f | g | o] on both sides: forward states AND every BPTT output against torch autograd;
  * SimpleRNN = torch.nn.RNNCell(tanh), exactly;
  * GRU   : Keras 2.0.x applies the reset gate BEFORE the candidate's recurrent matmul, (r * h) U_h, torch after it, r * (h U_h)
            (and orders the gates [r | z | n]).  The two coincide where U_h = 0 and where r = 1 - checked; with a general U_h
            they differ, and the difference is the closed form ((r * h) U_h - r * (h U_h)) inside the tanh - checked too;
  * Adam  : Keras 2.0.8 adds epsilon OUTSIDE the bias correction, torch inside; with epsilon_t = epsilon * sqrt(1 - beta_2^t)
            the oracle's update IS torch.optim.Adam's - so that placement is the only difference.
CPU only; float64."""
import numpy as np
import pytest
import torch

from oracle import vae_oracle as vo


def _problem(G, T=6, B=5, H=7, seed=0):
    rng = np.random.default_rng(seed)
    return (rng, rng.standard_normal((T, B, G * H)) * 0.7, rng.standard_normal((H, G * H)) * 0.4, rng.standard_normal((B, H)) * 0.5,
            rng.standard_normal((B, H)) * 0.5)


def _t(a, grad=False):
    return torch.tensor(a, dtype=torch.float64, requires_grad=grad)


def test_lstm_with_sigmoid_gates_is_torch_lstmcell_forward_and_backward():
    rng, xp, U, h0, c0 = _problem(4, seed=1)
    T, B, GH = xp.shape
    H = U.shape[0]
    hs, cs, acts = vo.rnn_forward("LSTM", xp, U, h0, c0, rec_act="sigmoid")
    cell = torch.nn.LSTMCell(GH, H, bias=False, dtype=torch.float64)      # x = xp itself through an identity input kernel
    with torch.no_grad():
        cell.weight_ih.copy_(torch.eye(GH, dtype=torch.float64))
    cell.weight_hh = torch.nn.Parameter(_t(U.T.copy()))
    xt, h0t, c0t = _t(xp, True), _t(h0, True), _t(c0, True)
    h, c, outs = h0t, c0t, []
    for t in range(T):
        h, c = cell(xt[t], (h, c))
        outs.append(h)
    hs_t = torch.stack(outs)
    assert np.allclose(hs_t.detach().numpy(), hs[1:], rtol=1e-12, atol=1e-13)
    assert np.allclose(c.detach().numpy(), cs[-1], rtol=1e-12, atol=1e-13)
    dext, dlast = rng.standard_normal((T, B, H)), rng.standard_normal((B, H))
    ((hs_t * _t(dext)).sum() + (hs_t[-1] * _t(dlast)).sum()).backward()
    da, dU, dh0, dc0 = vo.rnn_backward("LSTM", hs, cs, acts, U, dext, dlast, rec_act="sigmoid")
    assert np.allclose(da, xt.grad.numpy(), rtol=1e-10, atol=1e-12)
    assert np.allclose(dU, cell.weight_hh.grad.numpy().T, rtol=1e-10, atol=1e-12)
    assert np.allclose(dh0, h0t.grad.numpy(), rtol=1e-10, atol=1e-12)
    assert np.allclose(dc0, c0t.grad.numpy(), rtol=1e-10, atol=1e-12)
    # ... and the reference's hard_sigmoid is a different function of the same pre-activations: the switch is what made them equal
    assert not np.allclose(vo.rnn_forward("LSTM", xp, U, h0, c0)[0], hs, atol=1e-3)


def test_simple_rnn_is_torch_rnncell():
    rng, xp, U, h0, _ = _problem(1, seed=2)
    T, B, H = xp.shape
    hs, _, acts = vo.rnn_forward("SimpleRNN", xp, U, h0)
    cell = torch.nn.RNNCell(H, H, bias=False, nonlinearity="tanh", dtype=torch.float64)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.eye(H, dtype=torch.float64))
    cell.weight_hh = torch.nn.Parameter(_t(U.T.copy()))
    xt, h0t = _t(xp, True), _t(h0, True)
    h, outs = h0t, []
    for t in range(T):
        h = cell(xt[t], h)
        outs.append(h)
    hs_t = torch.stack(outs)
    assert np.allclose(hs_t.detach().numpy(), hs[1:], rtol=1e-12, atol=1e-13)
    dext = rng.standard_normal((T, B, H))
    (hs_t * _t(dext)).sum().backward()
    da, dU, dh0, _ = vo.rnn_backward("SimpleRNN", hs, None, acts, U, dext, None)
    assert np.allclose(da, xt.grad.numpy(), rtol=1e-10, atol=1e-12)
    assert np.allclose(dU, cell.weight_hh.grad.numpy().T, rtol=1e-10, atol=1e-12)
    assert np.allclose(dh0, h0t.grad.numpy(), rtol=1e-10, atol=1e-12)


def _torch_gru(xp, U, h0):
    """torch.nn.GRUCell on the oracle's operands: gate order [r | z | n] there, [z | r | h] here; reset AFTER the matmul there"""
    T, B, GH = xp.shape
    H = U.shape[0]
    perm = np.concatenate([np.arange(H, 2 * H), np.arange(0, H), np.arange(2 * H, 3 * H)])     # oracle column of torch gate row
    cell = torch.nn.GRUCell(GH, H, bias=False, dtype=torch.float64)
    eye = np.eye(GH)[perm]
    with torch.no_grad():
        cell.weight_ih.copy_(_t(eye))
    cell.weight_hh = torch.nn.Parameter(_t(U.T[perm].copy()))
    xt, h0t = _t(xp, True), _t(h0, True)
    h, outs = h0t, []
    for t in range(T):
        h = cell(xt[t], h)
        outs.append(h)
    return torch.stack(outs), xt, h0t, cell, perm


def test_gru_is_torch_grucell_where_reset_before_and_after_coincide():
    rng, xp, U, h0, _ = _problem(3, seed=3)
    T, B, GH = xp.shape
    H = U.shape[0]
    # (1) no recurrent candidate kernel: (r * h) 0 = r * (h 0)
    U0 = U.copy()
    U0[:, 2 * H:] = 0.0
    hs, _, acts = vo.rnn_forward("GRU", xp, U0, h0, rec_act="sigmoid")
    hs_t, xt, h0t, cell, perm = _torch_gru(xp, U0, h0)
    assert np.allclose(hs_t.detach().numpy(), hs[1:], rtol=1e-12, atol=1e-13)
    dext = rng.standard_normal((T, B, H))
    (hs_t * _t(dext)).sum().backward()
    da, dU, dh0, _ = vo.rnn_backward("GRU", hs, None, acts, U0, dext, None, rec_act="sigmoid")
    assert np.allclose(da, xt.grad.numpy(), rtol=1e-10, atol=1e-12)
    assert np.allclose(dh0, h0t.grad.numpy(), rtol=1e-10, atol=1e-12)
    g_hh = np.empty_like(U0)
    g_hh[:, perm] = cell.weight_hh.grad.numpy().T                  # torch gate rows back to oracle columns
    assert np.allclose(dU[:, :2 * H], g_hh[:, :2 * H], rtol=1e-10, atol=1e-12)        # (z, r kernels; the candidate's differs by form)
    # (2) reset gate saturated at 1 (pre-activation 40: sigmoid = 1 - 4e-18 in float64, i.e. 1): (1 * h) U = 1 * (h U)
    xp1 = xp.copy()
    xp1[:, :, H:2 * H] = 40.0
    hs1 = vo.rnn_forward("GRU", xp1, U * 0.05, h0, rec_act="sigmoid")[0]       # (small U: h U cannot pull r off saturation)
    hs1_t = _torch_gru(xp1, U * 0.05, h0)[0]
    assert np.allclose(hs1_t.detach().numpy(), hs1[1:], rtol=1e-12, atol=1e-13)


def test_gru_reset_before_differs_from_reset_after_by_the_closed_form():
    """one step, general U_h: Keras-2.0.x candidate tanh(x_h + (r*h) U_h) vs torch's tanh(x_h + r * (h U_h)); the oracle implements
    the first (the reference's: SURVEY A.2), torch the second, and the difference of the pre-activations is (r*h) U_h - r * (h U_h)"""
    rng, xp, U, h0, _ = _problem(3, T=1, seed=4)
    H = U.shape[0]
    hs, _, acts = vo.rnn_forward("GRU", xp, U, h0, rec_act="sigmoid")
    hs_t = _torch_gru(xp, U, h0)[0].detach().numpy()
    z, r, hh = acts[0, :, :H], acts[0, :, H:2 * H], acts[0, :, 2 * H:]
    pre_before = np.arctanh(hh)
    pre_after = pre_before - (r * h0) @ U[:, 2 * H:] + r * (h0 @ U[:, 2 * H:])
    want_torch = z * h0 + (1.0 - z) * np.tanh(pre_after)
    assert np.allclose(hs_t[0], want_torch, rtol=1e-10, atol=1e-12)
    assert np.abs(hs_t[0] - hs[1]).max() > 1e-3                    # ... and they really are different cells


@pytest.mark.parametrize("steps", [1, 7])
def test_keras_adam_is_torch_adam_up_to_where_epsilon_sits(steps):
    rng = np.random.default_rng(5)
    p0 = {"a": rng.standard_normal((4, 3)), "b": rng.standard_normal(5)}
    grads = [{k: rng.standard_normal(v.shape) * (10.0 ** rng.integers(-6, 1)) for k, v in p0.items()} for _ in range(steps)]
    lr, b1, b2, eps = 2e-4, 0.9, 0.999, 1e-8
    orc = vo.OracleVAE.__new__(vo.OracleVAE)
    orc.cfg = dict(lr=lr, optimizer="Adam")
    tp = {k: _t(v.copy(), True) for k, v in p0.items()}
    opt = torch.optim.Adam(list(tp.values()), lr=lr, betas=(b1, b2), eps=eps)
    pk, pe = {k: v.copy() for k, v in p0.items()}, {k: v.copy() for k, v in p0.items()}
    sk, se = orc.new_opt_state(pk), orc.new_opt_state(pe)
    for t, g in enumerate(grads, 1):
        for k in tp:
            tp[k].grad = _t(g[k])
        opt.step()
        orc.opt_step(pk, g, sk, b1, b2, eps)                                  # Keras 2.0.8: epsilon outside the bias correction
        orc.opt_step(pe, g, se, b1, b2, eps * np.sqrt(1.0 - b2 ** t))          # the same formula with torch's effective epsilon
    for k in tp:
        assert np.allclose(pe[k], tp[k].detach().numpy(), rtol=1e-13, atol=1e-15), k
        d = np.abs(pk[k] - tp[k].detach().numpy()).max()
        assert 0.0 < d < 2 * lr * steps, (k, d)                    # a real but bounded difference: gradients of ~1e-6 sit near epsilon
