"""The reference-held vectors on the HIP path (tests/golden/packers_decode.npz: produced by importing the reference's own NumPy
helpers, tests/golden/make_fixtures.py).

(a) the lists `prepare_autoencoder_input_and_output_list` returned for the reference (ae_x_* / ae_y_* / ae_w_*, default settings:
    T=64, latent 256, GRU) go through `autoencoder.evaluate`, one `autoencoder.fit` and `decoder.predict` ->
    `process_decoder_outputs('argmax')`, against the oracle on the same lists;
(b) the device argmax (mvae_head) against the reference's `sample_notes_prediction / sample_instrument_prediction('argmax')` of the
    golden probabilities, bit for bit - including the rows the generator planted: all zeros -> index 0, the silent class ->
    an all-zero output row, a tie -> the first maximum (reference vae_definition.py:1048-1067,1071-1107)."""
import numpy as np
import pytest
import torch

import midi_vae_amd  # noqa: F401
from midi_vae_amd import hiplib as hl, ops, packers as pk
from midi_vae_amd.config import build_settings, create_kwargs
from midi_vae_amd.layout import init_params
from midi_vae_amd.model import VAE
from oracle.vae_oracle import OracleVAE, make_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _device_argmax(probs2d, H=64):
    """rows of probabilities -> mvae_head with hs = log(p) through an identity weight: logits = log(p), the kernel's argmax"""
    R0, N = probs2d.shape
    R = (R0 + 15) // 16 * 16
    NP = ops.head_np(N)
    with np.errstate(divide="ignore"):
        logp = np.where(probs2d > 0, np.log(np.where(probs2d > 0, probs2d, 1.0)), -1e30)      # (a zero: far below, finite)
    hs = np.zeros((R, H), np.float32)
    hs[:R0, :N] = logp
    wt = np.zeros((NP, H), np.float32)
    wt[np.arange(N), np.arange(N)] = 1.0
    am = torch.full((R,), 255, dtype=torch.uint8, device=DEV)
    sc = torch.zeros((2,), device=DEV)
    ops.head(0, hl.F32, R, H, N, torch.tensor(hs, device=DEV), torch.tensor(wt, device=DEV), torch.zeros(N, device=DEV),
             argmax=am, scalars=sc)
    torch.cuda.synchronize()
    return am.cpu().numpy()[:R0]


def test_device_argmax_equals_the_references_decode_of_the_golden_probabilities(golden, default_settings):
    g, s = golden, default_settings
    pn, pi = g["probs_notes"], g["probs_instr"]
    idx = _device_argmax(pn.reshape(-1, pn.shape[-1]))
    assert idx.reshape(pn.shape[:2])[0, 3] == 0 and idx.reshape(pn.shape[:2])[1, 5] == 60 and idx.reshape(pn.shape[:2])[2, 7] == 10
    np.testing.assert_array_equal(pk.notes_from_indices(s, idx, pn.shape[-1]), g["notes_argmax_3d"])
    np.testing.assert_array_equal(pk.notes_from_indices(s, idx[:pn.shape[1]], pn.shape[-1]), g["notes_argmax_2d"])
    ii = _device_argmax(pi.reshape(-1, pi.shape[-1]))
    onehot = np.zeros(pi.shape)
    np.put_along_axis(onehot, ii.reshape(pi.shape[:2])[..., None].astype(np.int64), 1, axis=-1)
    np.testing.assert_array_equal(onehot, g["instr_argmax"])


def _lists(g):
    x = [g["ae_x_%d" % i] for i in range(int(g["ae_x_n"]))]
    y = [g["ae_y_%d" % i] for i in range(int(g["ae_y_n"]))]
    w = [g["ae_w_%d" % i] for i in range(int(g["ae_w_n"]))]
    return x, y, w


def test_golden_lists_through_evaluate_fit_and_decode_equal_the_oracle(golden, default_settings):
    g, s = golden, default_settings
    x, y, w = _lists(g)
    n = x[0].shape[0]
    m = VAE().create(compute_dtype="f32", seed=11, **create_kwargs(build_settings(epsilon_std=0.0)))
    spec = m.spec
    assert (spec.T, spec.Z, spec.cell) == (64, 256, "GRU")            # the reference's shipped settings
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    # Untrained, the decoder's cells relax to h = 0 and every softmax is uniform to 1e-8 (median top-2 gap 9e-9 measured): an
    # argmax comparison would test rounding noise.  Non-zero biases and larger output kernels make the outputs decisive.
    rng = np.random.default_rng(5)
    named = init_params(spec, 11)
    for k in named:
        if k.endswith(".b"):
            named[k] = (rng.standard_normal(named[k].shape) * 0.2).astype(np.float32)
        if k.endswith(".out.W"):
            named[k] = (named[k] * 8.0).astype(np.float32)
    m._shared.set_params(named)
    p = {k: v.astype(np.float64) for k, v in named.items()}
    batch = dict(X=x[0], Hist=x[2], I=x[4], Vel=x[6], Y=y[0], C=y[3], w_notes=w[0])
    eps = np.zeros((n, spec.Z))
    m_o, cache = orc.forward(p, batch, eps)
    # evaluate: [total, notes, instrument, velocity, style losses, accuracies...] (Keras metrics_names order)
    res = m.autoencoder.evaluate(x, y, batch_size=8, verbose=False)
    assert m.autoencoder.metrics_names[:5] == ["loss", "decoder_loss", "decoder_loss", "decoder_loss", "composer_decoder_loss"]
    for i, b in enumerate(("loss", "notes_loss", "instr_loss", "vel_loss", "style_loss")):
        assert abs(res[i] - m_o[b]) <= 2e-4 * (1 + abs(m_o[b])), (i, b, res[i], m_o[b])
    # decoder.predict on the lists' decoder inputs -> the reference's post-processing == the same applied to the oracle's outputs
    dec_in = [g["dec_in_autoH_%d" % i] for i in range(int(g["dec_in_n"]))]
    outs = m.decoder.predict(dec_in, batch_size=8)
    o_out = orc.decode(p, dec_in[1], dec_in[2], dict(notes=dec_in[0], instr=dec_in[3], vel=dec_in[4]))
    got = pk.process_decoder_outputs(s, outs, "argmax")
    want = pk.process_decoder_outputs(s, [o_out["notes"], o_out["instr"], o_out["vel"]], "argmax")
    for a, b, name in zip(got, want, "YIVDN"):
        if name == "V":
            np.testing.assert_allclose(a, b, atol=2e-5, err_msg=name)
        else:
            np.testing.assert_array_equal(a, b, err_msg=name)
    srt = np.sort(o_out["notes"], -1)
    assert (srt[..., -1] - srt[..., -2]).min() > 1e-5, "the oracle's outputs are not decisive enough to compare an argmax in f32"
    idx = m.decoder.predict_note_indices(dec_in, batch_size=8)
    np.testing.assert_array_equal(idx, np.argmax(o_out["notes"], -1))
    # one fit on the lists (Keras Adam) == the oracle's train step
    st = orc.new_opt_state(p)
    m_t = orc.train_step(p, st, batch, eps)
    h = m.autoencoder.fit(x, y, epochs=1, batch_size=8, shuffle=False, sample_weight=w, verbose=False)
    assert abs(h.history["loss"][0] - m_t["loss"]) <= 2e-4 * (1 + abs(m_t["loss"]))
    for name, a in zip(m.autoencoder._names(), m.autoencoder.get_weights()):
        assert np.allclose(a, p[name], rtol=2e-3, atol=3e-5), name
