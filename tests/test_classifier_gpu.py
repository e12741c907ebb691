"""The three style classifiers (reference pitch_/velocity_/instrument_classifier.py) on the engine vs oracle/classifier_oracle.py:
loss, accuracy, probabilities and every gradient (f32: 2e-4; bf16 on the resident H=256 kernels: 3e-2 / relative L2 < 6e-2), the
Keras-Adam trajectory of the per-song fit loop, and the Keras surface the scripts use."""
import numpy as np
import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd.classifier import ClassifierEngine, StyleClassifier
from midi_vae_amd.layout import ClassifierSpec, init_classifier_params
from oracle.classifier_oracle import OracleClassifier

pytestmark = pytest.mark.gpu


def _problem(xmode, B, T, K, C, H, L, seed, cell="GRU"):
    spec = ClassifierSpec(K=K, T=T, C=C, H=H, L=L, cell=cell, xmode=xmode, lr=1e-3)
    rng = np.random.default_rng(seed)
    params = init_classifier_params(spec, seed)
    for k in params:
        if k.endswith(".b"):
            params[k] = (rng.standard_normal(params[k].shape) * 0.1).astype(np.float32)
    if xmode == "index":
        x = rng.integers(0, K, (B, T)).astype(np.uint8)
        X = np.eye(K)[x.astype(np.int64)]
    else:
        x = np.where(rng.random((B, T)) < 0.5, 0.0, 0.5 + 0.5 * rng.random((B, T))).astype(np.float32)
        X = x[..., None].astype(np.float64)
    c = rng.integers(0, C, (B,)).astype(np.uint8)
    return spec, params, x, X, c, np.eye(C)[c.astype(np.int64)]


def _rel_l2(a, b):
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


@pytest.mark.parametrize("xmode,K,T", [("index", 61, 12), ("scalar", 1, 12), ("index", 16, 4)])
@pytest.mark.parametrize("L", [1, 2, 3])
@pytest.mark.parametrize("B", [5, 32])
def test_classifier_forward_backward_matches_oracle_f32(xmode, K, T, L, B):
    spec, params, x, X, c, Y = _problem(xmode, B, T, K, 3, 64, L, seed=L + B)
    orc = OracleClassifier(spec.oracle_cfg())
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    probs_o, m_o, cache = orc.forward(p64, X, Y)
    g_o = orc.backward(p64, cache)
    eng = ClassifierEngine(spec, max_batch=32, dtype="f32")
    eng.set_params(params)
    eng.stage(x, c)
    eng.grads.zero_()
    eng.forward(B, want_probs=True)
    eng.backward(B)
    m = eng.metrics(B)
    assert abs(m["loss"] - m_o["loss"]) <= 2e-4 * (1 + abs(m_o["loss"])) and abs(m["acc"] - m_o["acc"]) < 1e-9
    np.testing.assert_allclose(eng.probs(B), probs_o, rtol=2e-4, atol=2e-6)
    g = eng.get_grads()
    for k in g_o:
        err = np.abs(g[k] - g_o[k])
        assert np.all(err <= 2e-6 + 2e-4 * np.abs(g_o[k]) + 2e-4 * np.abs(g_o[k]).max()), (k, err.max())


@pytest.mark.parametrize("xmode,K,T", [("index", 61, 64), ("scalar", 1, 64), ("index", 16, 4)])
def test_classifier_resident_path_matches_oracle_bf16(xmode, K, T):
    """the reference's shape - 2 x GRU(256) - on the resident-weights bf16 kernels (time-pipelined for the 64-step rolls)"""
    B = 32
    spec, params, x, X, c, Y = _problem(xmode, B, T, K, 2, 256, 2, seed=7)
    orc = OracleClassifier(spec.oracle_cfg())
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    probs_o, m_o, cache = orc.forward(p64, X, Y)
    g_o = orc.backward(p64, cache)
    eng = ClassifierEngine(spec, max_batch=B, dtype="bf16")
    assert eng.tile16 and (eng._pipelined(eng.layers) == (T % eng.pipe_chunk == 0))
    eng.set_params(params)
    eng.stage(x, c)
    eng.grads.zero_()
    eng.forward(B, want_probs=True)
    eng.backward(B)
    eng.check_pipeline()
    m = eng.metrics(B)
    assert abs(m["loss"] - m_o["loss"]) <= 3e-2 * (1 + abs(m_o["loss"]))
    np.testing.assert_allclose(eng.probs(B), probs_o, rtol=3e-2, atol=3e-3)
    g = eng.get_grads()
    for k in g_o:
        assert _rel_l2(g[k], g_o[k]) < 6e-2, (k, _rel_l2(g[k], g_o[k]))


@pytest.mark.parametrize("kind", ["pitch", "velocity", "instrument"])
def test_per_song_fit_loop_matches_oracle_trajectory(kind):
    """model.fit per song (reference pitch_classifier.py:223-245; the instrument script feeds ONE (1, V, 16) sample per song and a
    bare one-hot target, instrument_classifier.py:231-244), then evaluate + predict + confusion matrix as its test() does."""
    rng = np.random.default_rng(11)
    C, H, L, bs = 2, 64, 2, 8
    if kind == "pitch":
        K, T, sizes = 61, 16, [13, 5, 8]
    elif kind == "velocity":
        K, T, sizes = 1, 16, [9, 12]
    else:
        K, T, sizes = 16, 4, [1, 1, 1]
    clf = StyleClassifier(kind, input_dim=K, num_classes=C, lstm_size=H, num_layers=L, learning_rate=1e-3, compute_dtype="f32", seed=4)
    spec = ClassifierSpec(K=K, T=T, C=C, H=H, L=L, xmode=clf.xmode, lr=1e-3)
    p = {k: v.astype(np.float64) for k, v in init_classifier_params(spec, 4).items()}
    orc = OracleClassifier(spec.oracle_cfg())
    st = orc.new_opt_state(p)
    songs = []
    for i, n in enumerate(sizes):
        if kind == "velocity":
            X = np.where(rng.random((n, T, 1)) < 0.5, 0.0, 0.5 + 0.5 * rng.random((n, T, 1)))
        else:
            X = np.eye(K)[rng.integers(0, K, (n, T))]
        cls = i % C
        Y = np.eye(C)[cls] if kind == "instrument" else np.tile(np.eye(C)[cls][None], (n, 1))
        songs.append((X, Y))
    for X, Y in songs:
        h = clf.fit(X, Y, epochs=1, batch_size=bs, shuffle=False, verbose=False)
        clf.reset_states()
        n = X.shape[0]
        Y2 = np.tile(np.asarray(Y)[None], (n, 1)) if np.asarray(Y).ndim == 1 else Y
        tot = acc = 0.0
        for lo in range(0, n, bs):
            hi = min(n, lo + bs)
            _, m, cache = orc.forward(p, X[lo:hi], Y2[lo:hi])
            orc.opt_step(p, orc.backward(p, cache), st, 1e-3)
            tot += m["loss"] * (hi - lo) / n
            acc += m["acc"] * (hi - lo) / n
        assert abs(h.history["loss"][0] - tot) < 1e-4 and abs(h.history["acc"][0] - acc) < 1e-9
    for a, k in zip(clf.get_weights(), p):
        assert np.allclose(a, p[k], rtol=2e-3, atol=3e-5), k
    X, Y = songs[0]
    Y2 = np.tile(np.asarray(Y)[None], (X.shape[0], 1)) if np.asarray(Y).ndim == 1 else Y
    loss, acc = clf.evaluate(X, Y, batch_size=bs, verbose=False)
    probs = clf.predict(X, batch_size=bs, verbose=False)
    probs_o, m_o, _ = orc.forward(p, X, Y2)
    assert abs(loss - m_o["loss"]) < 2e-4 and abs(acc - m_o["acc"]) < 1e-9
    np.testing.assert_allclose(probs, probs_o, rtol=2e-3, atol=2e-5)
    conf = np.zeros((C, C))
    for yv, yp in zip(Y2, probs):
        conf[np.argmax(yp), np.argmax(yv)] += 1
    assert conf.sum() == X.shape[0]
    if kind != "velocity":                     # rows that are not one-hot are refused, not silently arg-maxed
        with pytest.raises(NotImplementedError):
            clf.predict(np.full((2, T, K), 0.3), batch_size=bs)
