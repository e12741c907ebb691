"""The reference-shaped calls (fit / evaluate / predict on the packers' lists) end to end on the GPU vs the oracle."""
import numpy as np
import torch
import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd import packers as pk
from midi_vae_amd.config import build_settings, create_kwargs
from midi_vae_amd.model import VAE
from midi_vae_amd.synth import make_windows, to_reference_format
from oracle.vae_oracle import OracleVAE, history_from_z, make_cfg

pytestmark = pytest.mark.gpu


def _setup(cell, n=20, seed=0):
    s = build_settings(cell_type=cell, lstm_size=64, latent_dim=32, input_length=4, output_length=4, batch_size=8,
                       learning_rate=1e-3)
    m = VAE().create(compute_dtype="f32", seed=seed, **create_kwargs(s))
    w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=5)
    X, Y, C, I, V, D = to_reference_format(w)
    return s, m, (X, Y, C, I, V, D)


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_fit_history_matches_oracle_trajectory(cell):
    s, m, (X, Y, C, I, V, D) = _setup(cell)
    n, bs = X.shape[0], s["batch_size"]
    H = np.zeros((n, s["latent_dim"]))
    S = np.zeros((n, s["signature_vector_length"]))
    x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=True)
    hist = m.autoencoder.fit(x, y, epochs=1, batch_size=bs, shuffle=False, sample_weight=sw, verbose=False)
    for k in ("loss", "decoder_loss_1", "decoder_acc_1", "decoder_loss_2", "decoder_loss_3", "composer_decoder_loss",
              "composer_decoder_acc"):
        assert k in hist.history and len(hist.history[k]) == 1
    # oracle: same init, same minibatches (ragged last one: 20 = 8 + 8 + 4), same epsilon stream
    spec = m.spec
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    from midi_vae_amd.layout import init_params
    p = {k: v.astype(np.float64) for k, v in init_params(spec, 0).items()}
    st = orc.new_opt_state(p)
    rng = np.random.default_rng(1)
    tot = 0.0
    Coh = np.eye(s["num_classes"])[np.full(n, C)]
    It = np.tile(I[None], (n, 1, 1))
    for lo in range(0, n, bs):
        hi = min(n, lo + bs)
        eps = (rng.standard_normal((hi - lo, spec.Z)) * spec.epsilon_std).astype(np.float32).astype(np.float64)
        b = dict(X=X[lo:hi], I=It[lo:hi], Vel=V[lo:hi, :, None], Hist=H[lo:hi], Y=Y[lo:hi], C=Coh[lo:hi])
        tot += orc.train_step(p, st, b, eps)["loss"] * (hi - lo)
    assert abs(hist.history["loss"][0] - tot / n) < 1e-3
    got = m.autoencoder.get_weights()
    names = m.autoencoder._names()
    for nme, a in zip(names, got):
        assert np.allclose(a, p[nme], rtol=2e-3, atol=3e-5), nme


def test_evaluate_predict_and_history_prepass():
    s, m, (X, Y, C, I, V, D) = _setup("GRU")
    n = X.shape[0]
    enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
    z = m.encoder.predict(enc_in, batch_size=s["batch_size"], verbose=False)
    assert z.shape == (n, s["latent_dim"]) and np.all(np.isfinite(z))
    H = history_from_z(z)                                  # reference vae_training.py:795-798
    S = np.zeros((n, s["signature_vector_length"]))
    x, y = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H)
    res = m.autoencoder.evaluate(x, y, batch_size=s["batch_size"], verbose=False)
    names = m.autoencoder.metrics_names
    assert len(res) == len(names) == 9
    total = res[0]
    parts = res[1] + s["meta_instrument_weight"] * res[2] + s["meta_velocity_weight"] * res[3] + s["composer_weight"] * res[4]
    kl = total - parts                                     # the reference derives KL this way (vae_training.py:525-538)
    assert 0 <= kl < 1.0
    outs = m.autoencoder.predict(x, batch_size=s["batch_size"])
    assert [o.shape for o in outs] == [(n, 16, 61), (n, 4, 16), (n, 16, 1), (n, 2)]
    assert np.allclose(outs[0].sum(-1), 1, atol=1e-5)
    # decoder alone with a latent swap (reference vae_evaluation.py:2471-2483), then the host argmax decode
    z2 = z.copy()
    z2[:, [0, 1]] = z2[:, [1, 0]]
    dec_in = pk.prepare_decoder_input(s, z2, C, S, None)
    d_out = m.decoder.predict(dec_in, batch_size=s["batch_size"])
    Yd, Id, Vd, Dd, Nd = pk.process_decoder_outputs(s, d_out, "argmax")
    assert Yd.shape == (n * 16, 60)
    idx = m.decoder.predict_note_indices(dec_in, batch_size=s["batch_size"])
    assert np.array_equal(pk.notes_from_indices(s, idx, 61), Yd)      # fused device argmax == host argmax decode


def test_weights_survive_engine_regrowth(tmp_path):
    s, m, (X, Y, C, I, V, D) = _setup("GRU")
    enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
    m._shared.rng = np.random.default_rng(0)
    z1 = m.encoder.predict(enc_in, batch_size=4)
    m._shared.rng = np.random.default_rng(0)
    z2 = m.encoder.predict(enc_in, batch_size=20)          # bigger batch -> engine is rebuilt, weights carried over
    assert np.allclose(z1[:4], z2[:4], atol=1e-5)


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_held_and_next_heads_through_the_reference_lists(cell):
    """meta_held_notes + meta_next_notes (+ the teacher-forcing switches, which add inputs but do not reach the cell graph -
    SURVEY F9) through the reference's own list layout (vae_definition.py:880-1045: the next-notes target is the NEXT window,
    the last window is dropped): evaluate equals the oracle's forward pass on the same lists, fit reports
    decoder_loss_1..5 / decoder_acc_1..5, predict returns the five decoder outputs + style."""
    s = build_settings(cell_type=cell, lstm_size=64, latent_dim=32, input_length=4, output_length=4, batch_size=8,
                       learning_rate=1e-3, meta_held_notes=True, meta_next_notes=True, teacher_force=True,
                       meta_next_notes_teacher_force=False, epsilon_std=0.0)
    m = VAE().create(compute_dtype="f32", seed=2, **create_kwargs(s))
    n = 13
    w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=8)
    X, Y, C, I, V, D = to_reference_format(w)
    rng = np.random.default_rng(3)
    D = (rng.random(D.shape) < 0.4).astype(np.float64)
    Hh = rng.standard_normal((n, s["latent_dim"])) * 0.1
    S = np.zeros((n, s["signature_vector_length"]))
    x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, Hh, return_sample_weight=True)
    assert x[0].shape[0] == n - 1 and len(y) == 6 and len(sw) == 6
    names = m.autoencoder.metrics_names
    assert names.count("decoder_loss") == 5 and names.count("decoder_acc") == 5
    res = dict(zip(["loss"] + ["l%d" % i for i in range(1, 6)] + ["ls"] + ["a%d" % i for i in range(1, 6)] + ["as"],
                   m.autoencoder.evaluate(x, y, batch_size=s["batch_size"], verbose=False)))
    # the oracle on the same lists (epsilon_std = 0: z = z_mean, no noise to reproduce)
    spec = m.spec
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    from midi_vae_amd.layout import init_params
    p = {k: v.astype(np.float64) for k, v in init_params(spec, 2).items()}
    nb = n - 1
    tot = {}
    for lo in range(0, nb, s["batch_size"]):
        hi = min(nb, lo + s["batch_size"])
        b = dict(X=x[0][lo:hi], Hist=x[3][lo:hi], I=x[5][lo:hi], Vel=x[7][lo:hi], Held=x[9][lo:hi],
                 Y=y[0][lo:hi], Next=y[4][lo:hi], C=y[5][lo:hi])
        mo, _ = orc.forward(p, b, np.zeros((hi - lo, spec.Z)))
        for k, v in mo.items():
            tot[k] = tot.get(k, 0.0) + v * (hi - lo) / nb
    for got, want in (("loss", "loss"), ("l1", "notes_loss"), ("l2", "instr_loss"), ("l3", "vel_loss"), ("l4", "held_loss"),
                      ("l5", "next_loss"), ("ls", "style_loss"), ("a4", "held_acc"), ("a5", "next_acc")):
        assert abs(res[got] - tot[want]) <= 2e-4 * (1 + abs(tot[want])), (got, res[got], tot[want])
    hist = m.autoencoder.fit(x, y, epochs=2, batch_size=s["batch_size"], shuffle=False, sample_weight=sw, verbose=False)
    for k in ("decoder_loss_4", "decoder_acc_4", "decoder_loss_5", "decoder_acc_5"):
        assert len(hist.history[k]) == 2
    assert hist.history["loss"][1] < hist.history["loss"][0]
    outs = m.autoencoder.predict(x, batch_size=s["batch_size"])
    assert [o.shape for o in outs] == [(nb, 16, 61), (nb, 4, 16), (nb, 16, 1), (nb, 16, 2), (nb, 16, 61), (nb, 2)]
    enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
    assert len(enc_in) == 4
    z = m.encoder.predict(enc_in, batch_size=s["batch_size"])
    dec_in = pk.prepare_decoder_input(s, z, C, S, None)
    d_out = m.decoder.predict(dec_in, batch_size=s["batch_size"])
    assert [o.shape for o in d_out] == [(n, 16, 61), (n, 4, 16), (n, 16, 1), (n, 16, 2), (n, 16, 61)]


def test_signature_and_output_classifier_heads_through_the_reference_lists():
    """signature_decoder, composer_decoder_at_notes_output / _at_instrument_output and the decoder's additional input
    (append_signature_vector_to_latent + decoder_input_composer) through the reference's list layout (vae_definition.py:880-1045):
    evaluate equals the oracle's forward pass, fit reports the reference's history keys (vae_training.py:817-864), predict returns
    the outputs in the reference's order."""
    s = build_settings(cell_type="GRU", lstm_size=64, latent_dim=32, input_length=4, output_length=4, batch_size=8,
                       learning_rate=1e-3, signature_decoder=True, composer_decoder_at_notes_output=True,
                       composer_decoder_at_instrument_output=True, append_signature_vector_to_latent=True,
                       decoder_input_composer=True, epsilon_std=0.0)
    m = VAE().create(compute_dtype="f32", seed=2, **create_kwargs(s))
    n = 11
    w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=8)
    X, Y, C, I, V, D = to_reference_format(w)
    rng = np.random.default_rng(3)
    Hh = rng.standard_normal((n, s["latent_dim"])) * 0.1
    S = rng.standard_normal((n, s["signature_vector_length"])) * 0.4
    x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, Hh, return_sample_weight=True)
    assert len(x) == 8 and len(y) == 7 and len(sw) == 7 and x[3].shape == (n, 17)
    names = m.autoencoder.metrics_names
    assert names[4:8] == ["composer_decoder_loss", "signature_decoder_loss", "composer_decoder_at_notes_loss",
                          "composer_decoder_at_instruments_loss"]
    res = dict(zip(["loss", "l1", "l2", "l3", "ls", "lsig", "lcn", "lci", "a1", "a2", "a3", "as", "asig", "acn", "aci"],
                   m.autoencoder.evaluate(x, y, batch_size=s["batch_size"], verbose=False)))
    spec = m.spec
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    from midi_vae_amd.layout import init_params
    p = {k: v.astype(np.float64) for k, v in init_params(spec, 2).items()}
    tot = {}
    for lo in range(0, n, s["batch_size"]):
        hi = min(n, lo + s["batch_size"])
        b = dict(X=x[0][lo:hi], Hist=x[2][lo:hi], Add=x[3][lo:hi], I=x[5][lo:hi], Vel=x[7][lo:hi], Y=y[0][lo:hi], C=y[3][lo:hi],
                 S=y[4][lo:hi])
        mo, _ = orc.forward(p, b, np.zeros((hi - lo, spec.Z)))
        for k, v in mo.items():
            tot[k] = tot.get(k, 0.0) + v * (hi - lo) / n
    for got, want in (("loss", "loss"), ("l1", "notes_loss"), ("ls", "style_loss"), ("lsig", "sig_loss"), ("lcn", "cnotes_loss"),
                      ("lci", "cinstr_loss"), ("asig", "sig_acc"), ("acn", "cnotes_acc"), ("aci", "cinstr_acc")):
        assert abs(res[got] - tot[want]) <= 2e-4 * (1 + abs(tot[want])), (got, res[got], tot[want])
    hist = m.autoencoder.fit(x, y, epochs=2, batch_size=s["batch_size"], shuffle=False, sample_weight=sw, verbose=False)
    for k in ("signature_decoder_loss", "signature_decoder_acc", "composer_decoder_at_notes_loss", "composer_decoder_at_notes_acc",
              "composer_decoder_at_instruments_loss", "composer_decoder_at_instruments_acc"):
        assert len(hist.history[k]) == 2
    assert hist.history["loss"][1] < hist.history["loss"][0]
    outs = m.autoencoder.predict(x, batch_size=s["batch_size"])
    assert [o.shape for o in outs] == [(n, 16, 61), (n, 4, 16), (n, 16, 1), (n, 2), (n, 15), (n, 2), (n, 2)]


def test_device_history_equals_host_history():
    """f-1: the history pre-pass kept in HBM.  ``encoder.predict(device=True)`` returns a DeviceLatent; in the history slot of the
    fit / evaluate lists it stands for the rolled array the reference builds on the host (vae_training.py:795-798: H[1:] = z[:-1],
    H[0] = 0).  Same epsilon stream on both sides -> same z, same history, same losses after a fit call (ragged minibatches:
    the roll crosses minibatch boundaries) and the same parameters."""
    from midi_vae_amd.model import DeviceLatent
    res = {}
    for on_device in (True, False):
        s, m, (X, Y, C, I, V, D) = _setup("GRU", n=21, seed=4)
        n = X.shape[0]
        S = np.zeros((n, s["signature_vector_length"]))
        m._shared.rng = np.random.default_rng(7)
        enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
        z = m.encoder.predict(enc_in, batch_size=s["batch_size"], verbose=False, device=on_device)
        if on_device:
            assert isinstance(z, DeviceLatent) and z.shape == (n, s["latent_dim"])
            H = z
            res["z"] = z.latent()
            np.testing.assert_array_equal(z.numpy()[1:], res["z"][:-1])
            assert not z.numpy()[0].any()
        else:
            np.testing.assert_allclose(z, res["z"], rtol=0, atol=1e-6)
            H = history_from_z(z)
        x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=True)
        ev = m.autoencoder.evaluate(x, y, batch_size=s["batch_size"], verbose=False)
        hist = m.autoencoder.fit(x, y, epochs=1, batch_size=s["batch_size"], shuffle=False, sample_weight=sw, verbose=False)
        res[on_device] = (ev, {k: v[0] for k, v in hist.history.items()}, m.autoencoder.get_weights())
    (e1, h1, w1), (e0, h0, w0) = res[True], res[False]
    # (evaluate / fit draw their own epsilon: the same stream on both sides because both sides made the same calls before)
    np.testing.assert_allclose(e1, e0, rtol=1e-5, atol=1e-6)
    for k in h0:
        assert abs(h1[k] - h0[k]) <= 1e-5 * (1 + abs(h0[k])), (k, h1[k], h0[k])
    for a, b in zip(w1, w0):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)


@pytest.mark.parametrize("prepass", [False, True])
def test_fit_with_the_next_minibatch_converted_ahead_equals_serial_staging(prepass):
    """Autoencoder.prefetch (round 5): minibatch i+1 is converted by the host packers on a worker thread while step i is enqueued
    (Stager.prefetch; reference counterpart: Keras slices the next batch inside fit, vae_training.py:804-809).  Same epsilon
    stream (one draw per minibatch, in minibatch order), same mirrors, same uploads: two epochs over ragged minibatches
    (21 = 8 + 8 + 5) give the losses and the parameters of a fit that converts every minibatch on the caller's thread."""
    res = {}
    for ahead in (True, False):
        s, m, (X, Y, C, I, V, D) = _setup("GRU", n=21, seed=4)
        m.autoencoder.prefetch = ahead
        n = X.shape[0]
        S = np.zeros((n, s["signature_vector_length"]))
        m._shared.rng = np.random.default_rng(7)
        H = (m.encoder.predict(pk.prepare_encoder_input_list(s, X, I, V, D), batch_size=s["batch_size"], verbose=False, device=True)
             if prepass else np.zeros((n, s["latent_dim"])))
        x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=True)
        hist = m.autoencoder.fit(x, y, epochs=2, batch_size=s["batch_size"], shuffle=False, sample_weight=sw, verbose=False)
        st = m._shared.engine.stager()
        assert (st._worker is not None) == ahead
        res[ahead] = ({k: list(v) for k, v in hist.history.items()}, m.autoencoder.get_weights())
    (h1, w1), (h0, w0) = res[True], res[False]
    for k in h0:
        np.testing.assert_allclose(h1[k], h0[k], rtol=1e-5, atol=1e-6, err_msg=k)
    for a, b in zip(w1, w0):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)


def test_a_prefetched_conversion_is_taken_only_for_the_very_same_arrays():
    """Stager.stage(prefetched=True) used to accept a prefetched conversion when only the mirror and the window range
    matched.  Another song of equal length (or an epsilon drawn again) behind a prefetch of the first one must be converted
    afresh: the engine's input block then equals the one a plain stage() of the second song uploads."""
    from midi_vae_amd.staging import Norm
    s, m, (X, Y, C, I, V, D) = _setup("GRU", n=8, seed=4)
    _, _, (X2, Y2, C2, I2, V2, D2) = _setup("GRU", n=8, seed=4)
    rng = np.random.default_rng(11)
    perm = rng.permutation(8)
    X2, Y2, C2, I2, V2, D2 = (a[perm] if isinstance(a, np.ndarray) and a.shape[:1] == (8,) else a for a in (X2, Y2, C2, I2, V2, D2))
    n, ae = 8, m.autoencoder
    S, H = np.zeros((n, s["signature_vector_length"])), np.zeros((n, s["latent_dim"]))
    eng = m._shared.get_engine(s["batch_size"], training=True)
    st = eng.stager()

    def kw_of(X, Y, C, I, V, D, eps):
        x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=True)
        a = ae._unpack_x(x)
        a.update(ae._unpack_y(y))
        ws = ae._unpack_w(sw, n)
        a.update(ws)
        a["hist"], a["hist_dev"] = ae._history_source(a.get("hist"))
        return dict(eps=eps, norm=Norm.of(0, n, m.spec.T, **ws), **a)

    eps = rng.standard_normal((n, s["latent_dim"])).astype(np.float32) * 0.01
    kw1, kw2 = kw_of(X, Y, C, I, V, D, eps), kw_of(X2, Y2, C2, I2, V2, D2, eps)
    st.stage(0, n, **kw2)
    torch.cuda.synchronize()
    want = eng._in_block.clone()
    st.stage(0, n, **kw1)                        # (the other mirror: the next call uses the first one again)
    st.prefetch(0, n, **kw1)
    st.stage(0, n, prefetched=True, **kw2)       # the same window range, other arrays
    torch.cuda.synchronize()
    assert torch.equal(eng._in_block, want)
    st.prefetch(0, n, **kw1)
    st.stage(0, n, prefetched=True, **kw1)       # ... and the prefetched arrays themselves are taken
    torch.cuda.synchronize()
    assert not torch.equal(eng._in_block, want)


def test_history_is_filled_on_first_access_and_survives_later_fit_calls():
    """History.history is read from the device on first access (model.History): histories of several fit calls read AFTER the
    last call hold the same values as histories read immediately after each call."""
    vals = {}
    for lazy in (False, True):
        s, m, (X, Y, C, I, V, D) = _setup("GRU", n=19, seed=6)
        n = X.shape[0]
        S, H = np.zeros((n, s["signature_vector_length"])), np.zeros((n, s["latent_dim"]))
        x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=True)
        hs, got = [], []
        for _ in range(3):
            h = m.autoencoder.fit(x, y, epochs=2, batch_size=s["batch_size"], shuffle=False, sample_weight=sw, verbose=False)
            assert h.epoch == [0, 1]
            if lazy:
                hs.append(h)
                assert h._resolve is not None
            else:
                got.append({k: list(v) for k, v in h.history.items()})
        if lazy:
            got = [{k: list(v) for k, v in h.history.items()} for h in hs]
            assert all(h._resolve is None for h in hs)
        vals[lazy] = got
    for a, b in zip(vals[True], vals[False]):
        assert a.keys() == b.keys()
        for k in a:
            assert len(a[k]) == 2 and np.allclose(a[k], b[k], rtol=1e-6, atol=1e-7), (k, a[k], b[k])
    assert vals[False][2]["loss"][1] < vals[False][0]["loss"][0]


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
@pytest.mark.parametrize("n,bs", [(7, 8), (21, 8)])
def test_fused_history_prepass_matches_the_oracle(cell, n, bs):
    """f-1, checked against the ORACLE (not against the host path): reference vae_training.py:788-809 runs
    ``z' = encoder.predict(song)`` (fresh draw eps'), rolls it into the history input and then ``fit`` - both on the weights
    the fit call starts from.  Here ``encoder.predict(device=True)`` returns a deferred DeviceLatent and ``fit`` takes the first
    minibatch's history out of that minibatch's own encoder forward (one encoder pass for a one-minibatch song: n=7) and encodes
    the windows of the later minibatches forward-only before its first update (n=21: 8 + 8 + 5).  The oracle does what the
    reference does: ``encode`` with the PRE-step parameters and eps' gives z'; its train steps on H = roll(z') with the fit
    call's draws give the history's loss, and the parameters after the call."""
    from midi_vae_amd.layout import init_params
    from midi_vae_amd.model import DeviceLatent
    s = build_settings(cell_type=cell, lstm_size=64, latent_dim=32, input_length=4, output_length=4, batch_size=bs,
                       learning_rate=1e-3, epsilon_std=0.7)
    m = VAE().create(compute_dtype="f32", seed=3, **create_kwargs(s))
    w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=15)
    X, Y, C, I, V, D = to_reference_format(w)
    S = np.zeros((n, s["signature_vector_length"]))
    m._shared.rng = np.random.default_rng(11)
    enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
    lat = m.encoder.predict(enc_in, batch_size=bs, verbose=False, device=True)
    assert isinstance(lat, DeviceLatent) and lat.deferred and lat.shape == (n, s["latent_dim"])
    x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, lat, return_sample_weight=True)
    hist = m.autoencoder.fit(x, y, epochs=1, batch_size=bs, shuffle=False, sample_weight=sw, verbose=False)
    assert not lat.deferred
    if n <= bs:
        assert m._shared.infer is None, "a one-minibatch song must not run a separate encoder pass"
    # ---- the oracle, doing what the reference does
    spec = m.spec
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p = {k: v.astype(np.float64) for k, v in init_params(spec, 3).items()}
    rng = np.random.default_rng(11)
    draw = lambda k: (rng.standard_normal((k, spec.Z)) * spec.epsilon_std).astype(np.float32).astype(np.float64)
    eps2 = np.concatenate([draw(min(n, lo + bs) - lo) for lo in range(0, n, bs)], 0)       # encoder.predict: one draw per batch
    It = np.tile(I[None], (n, 1, 1))
    z_pre = orc.encode(p, X, It, V[:, :, None], eps2)
    np.testing.assert_allclose(lat.latent(), z_pre, rtol=0, atol=2e-5)
    H = history_from_z(z_pre)
    st = orc.new_opt_state(p)
    Coh = np.eye(s["num_classes"])[np.full(n, C)]
    tot = {}
    for lo in range(0, n, bs):
        hi = min(n, lo + bs)
        b = dict(X=X[lo:hi], I=It[lo:hi], Vel=V[lo:hi, :, None], Hist=H[lo:hi], Y=Y[lo:hi], C=Coh[lo:hi])
        for k, v in orc.train_step(p, st, b, draw(hi - lo)).items():
            tot[k] = tot.get(k, 0.0) + v * (hi - lo) / n
    h = {k: v[0] for k, v in hist.history.items()}
    for got, want in (("loss", "loss"), ("decoder_loss_1", "notes_loss"), ("decoder_loss_2", "instr_loss"),
                      ("decoder_loss_3", "vel_loss"), ("composer_decoder_loss", "style_loss")):
        assert abs(h[got] - tot[want]) <= 2e-4 * (1 + abs(tot[want])), (got, h[got], tot[want])
    for nme, a in zip(m.autoencoder._names(), m.autoencoder.get_weights()):
        assert np.allclose(a, p[nme], rtol=2e-3, atol=3e-5), nme


def test_deferred_latent_refuses_changed_weights():
    """the pre-pass must see the weights fit starts from: a deferred DeviceLatent used after an update raises"""
    s, m, (X, Y, C, I, V, D) = _setup("GRU", n=9)
    n = X.shape[0]
    S = np.zeros((n, s["signature_vector_length"]))
    enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
    lat = m.encoder.predict(enc_in, batch_size=8, device=True)
    x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, np.zeros((n, s["latent_dim"])),
                                                            return_sample_weight=True)
    m.autoencoder.fit(x, y, epochs=1, batch_size=8, shuffle=False, sample_weight=sw, verbose=False)
    with pytest.raises(RuntimeError):
        lat.latent()


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_chip_filling_internal_batch_gives_the_callers_results_h256_bf16(cell, monkeypatch):
    """encoder.predict / evaluate / decoder.predict_note_indices run at the forward-only engine's batch, whatever ``batch_size``
    the caller passes (reference vae_training.py:289,300, vae_evaluation.py:2482: Keras loops over batch_size windows).  The
    results are per window and epsilon is drawn per caller batch, so a model capped at the caller's batch (MVAE_INFER_BATCH=32:
    time-pipelined stacks, 32 windows at a time) and one that takes all 400 windows at once (the layers as chunk launches: the
    stack's kernels would not all be resident) must agree."""
    n, bs = 400, 32
    res = {}
    for cap in (32, 2048):
        monkeypatch.setenv("MVAE_INFER_BATCH", str(cap))
        s = build_settings(cell_type=cell, lstm_size=256, latent_dim=64, input_length=16, output_length=16, batch_size=bs,
                           epsilon_std=0.5)
        m = VAE().create(compute_dtype="bf16", seed=5, **create_kwargs(s))
        w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=21)
        X, Y, C, I, V, D = to_reference_format(w)
        S = np.zeros((n, s["signature_vector_length"]))
        m._shared.rng = np.random.default_rng(2)
        enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
        z = m.encoder.predict(enc_in, batch_size=bs)
        eng = m._shared.infer
        assert eng.maxB == (32 if cap == 32 else 512)
        x, y = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, history_from_z(z))
        ev = m.autoencoder.evaluate(x, y, batch_size=bs)
        idx = m.decoder.predict_note_indices(pk.prepare_decoder_input(s, z, C, S, None), batch_size=bs)
        res[cap] = (z, ev, idx)
    (z0, e0, i0), (z1, e1, i1) = res[32], res[2048]
    np.testing.assert_allclose(z1, z0, rtol=0, atol=1e-5)          # (same kernels, same arithmetic per window)
    np.testing.assert_allclose(e1, e0, rtol=2e-4, atol=1e-5)       # (sums over another partition of the windows)
    assert np.mean(i1 == i0) > 0.999


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_two_hundred_alternating_calls_never_time_out(cell):
    """200 alternating fit / encoder.predict / evaluate / decoder.predict_note_indices calls on ONE model (H=256
    bf16: training engine on time-pipelined stacks + K-streaming gradients, forward-only engine beside it on the same streams,
    songs of 5 .. 70 windows so that ragged batches, fused and separate history pre-passes and both engines alternate): no
    fallback warning, no time-out status on either engine, finite results throughout."""
    import warnings
    s = build_settings(cell_type=cell, lstm_size=256, latent_dim=64, input_length=16, output_length=16, batch_size=32,
                       learning_rate=2e-4)
    m = VAE().create(compute_dtype="bf16", seed=9, **create_kwargs(s))
    rng = np.random.default_rng(0)
    songs = []
    for i, n in enumerate((5, 32, 33, 70, 17)):
        w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=40 + i)
        songs.append(to_reference_format(w) + (np.zeros((n, s["signature_vector_length"])),))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        calls = 0
        while calls < 200:
            X, Y, C, I, V, D, S = songs[int(rng.integers(len(songs)))]
            n = X.shape[0]
            enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
            lat = m.encoder.predict(enc_in, batch_size=32, device=bool(calls % 3))       # deferred latent, or a host array
            H = lat if calls % 3 else history_from_z(lat)
            calls += 1
            if calls % 2:
                x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=True)
                h = m.autoencoder.fit(x, y, epochs=1, batch_size=32, shuffle=False, sample_weight=sw, verbose=False)
                assert np.isfinite(h.history["loss"][0])
            else:
                x, y = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H)
                assert np.all(np.isfinite(m.autoencoder.evaluate(x, y, batch_size=32, verbose=False)))
            calls += 1
            if calls % 5 == 0:
                z = lat.latent() if calls % 3 and hasattr(lat, "latent") else np.asarray(rng.standard_normal((n, s["latent_dim"])))
                idx = m.decoder.predict_note_indices(pk.prepare_decoder_input(s, z, C, S, None), batch_size=32)
                assert idx.shape == (n, 64) and idx.max() < 61
                calls += 1
    assert not [str(w.message) for w in rec if "timed out" in str(w.message)], [str(w.message) for w in rec]
    for eng in (m._shared.engine, m._shared.infer):
        eng.check_pipeline()
        assert eng.pipeline, "the engine fell back to chunked launches"


def test_attach_instruments_through_the_reference_lists():
    """the last live settings.py switch (attach_instruments=True, instrument_attach_method='1hot-category'): two-hot float64 rows
    (reference import_midi.py:288-292) through the host packers (mvae_host_twohot_to_index_tm), fit / evaluate / predict on the
    reference's lists; evaluate equals the oracle's forward pass on the same lists, the loss falls over two epochs."""
    from midi_vae_amd.layout import init_params
    s = build_settings(cell_type="GRU", lstm_size=64, latent_dim=32, input_length=4, output_length=4, batch_size=8,
                       learning_rate=1e-3, attach_instruments=True, epsilon_std=0.0)
    assert s["input_dim"] == s["output_dim"] == 77 and s["instrument_dim"] == 16
    m = VAE().create(compute_dtype="f32", seed=2, attach_dim=s["instrument_dim"], **create_kwargs(s))
    n = 13
    w = make_windows(n, s["output_length"], 61, s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=8)
    X0, _, C, I, V, D = to_reference_format(w)
    T = X0.shape[1]
    inst = np.tile(I, (T // s["max_voices"], 1))                       # (T, 16): row t = voice t % V (reference import_midi.py:291)
    X = np.concatenate([X0, np.tile(inst[None], (n, 1, 1))], -1)
    Y = X.copy()
    Hh = np.random.default_rng(3).standard_normal((n, s["latent_dim"])) * 0.1
    S = np.zeros((n, s["signature_vector_length"]))
    x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, Hh, return_sample_weight=True)
    res = m.autoencoder.evaluate(x, y, batch_size=8, verbose=False)
    orc = OracleVAE(make_cfg(**m.spec.oracle_cfg()))
    p = {k: v.astype(np.float64) for k, v in init_params(m.spec, 2).items()}
    tot = {}
    for lo in range(0, n, 8):
        hi = min(n, lo + 8)
        b = dict(X=x[0][lo:hi], Hist=x[2][lo:hi], I=x[4][lo:hi], Vel=x[6][lo:hi], Y=y[0][lo:hi], C=y[3][lo:hi])
        mo, _ = orc.forward(p, b, np.zeros((hi - lo, m.spec.Z)))
        for k, v in mo.items():
            tot[k] = tot.get(k, 0.0) + v * (hi - lo) / n
    names = m.autoencoder.metrics_names
    got = dict(zip(["loss", "l1", "l2", "l3", "ls", "a1", "a2", "a3", "as"], res))
    assert len(names) == 9
    for gk, ok in (("loss", "loss"), ("l1", "notes_loss"), ("a1", "notes_acc"), ("l2", "instr_loss"), ("ls", "style_loss")):
        assert abs(got[gk] - tot[ok]) <= 2e-4 * (1 + abs(tot[ok])), (gk, got[gk], tot[ok])
    hist = m.autoencoder.fit(x, y, epochs=2, batch_size=8, shuffle=False, sample_weight=sw, verbose=False)
    assert hist.history["loss"][1] < hist.history["loss"][0]
    outs = m.autoencoder.predict(x, batch_size=8)
    assert outs[0].shape == (n, T, 77) and np.allclose(outs[0].sum(-1), 1, atol=1e-5)
    with pytest.raises(NotImplementedError):           # a model built without attach_dim names the switch when it meets such rows
        s0 = build_settings(cell_type="GRU", lstm_size=64, latent_dim=32, input_length=4, output_length=4, batch_size=8,
                            attach_instruments=True)
        VAE().create(compute_dtype="f32", seed=2, **create_kwargs(s0)).autoencoder.evaluate(x, y, batch_size=8, verbose=False)
