"""The benched kernels and every BASELINE.json configuration under parity tests (GPU box only).

What runs in ``bench.py`` is the H=256 bf16 path: resident-weights slot-interleaved recurrent kernels, time-pipelined stacks,
fused latent chain, fast GEMMs.  These tests compare THAT path with the float64 oracle tensor by tensor (not only its loss),
at the sequence length the bench runs (T=512) and at the lengths of BASELINE configs[2..4] (T=2048, T=4096), and check
the full-size shapes of those configurations through size-independent properties:

  * shard linearity: every loss of this graph is a batch mean and batch rows are independent, so the gradient of the full
    minibatch equals the mean of the gradients of its 16-row shards, and the loss the mean of theirs - which is also
    the data-parallel contract;
  * row independence: encoder z / decoder argmax of rows taken from the far end of a full-size batch equal the same rows run
    as a 16-row batch (addresses beyond 2 GiB, 64 pipeline chunks per stack);
  * determinism, index range, falling loss, no pipeline time-out.

Tolerances: bf16 mode 3e-2 on losses, relative L2 < 6e-2 per gradient tensor against the oracle; argmax bit-exact (f32
mode against the oracle up to exact near-ties, stated where used).
"""
import numpy as np
import pytest
import torch

import midi_vae_amd  # noqa: F401
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec, init_params
from midi_vae_amd.synth import make_windows
from oracle import vae_oracle as vo
from oracle.vae_oracle import OracleVAE, make_cfg

pytestmark = pytest.mark.gpu


def _oh(idx, n):
    return np.eye(n)[idx.astype(np.int64)]


def _rel_l2(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-30)


def _setup(cell, B, T, V, Z, C, seed, hist_scale=0.1):
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=C, Le=2, Ld=2)
    rng = np.random.default_rng(seed)
    params = init_params(spec, seed)
    for k in params:                       # non-zero biases so every path carries signal
        if k.endswith(".b"):
            params[k] = (params[k] + rng.standard_normal(params[k].shape) * 0.05).astype(np.float32)
    w = make_windows(B, T, 61, V, 16, C, Z, seed=seed + 1, epsilon_std=spec.epsilon_std)
    w["hist"] = (rng.standard_normal((B, Z)) * hist_scale).astype(np.float32)
    batch = dict(X=_oh(w["x_idx"], 61), I=_oh(w["i_idx"], 16), Vel=w["vel"][..., None].astype(np.float64),
                 Hist=w["hist"].astype(np.float64), Y=_oh(w["x_idx"], 61), C=_oh(w["c_idx"], C))
    return spec, params, w, batch


def _stage(eng, w, B, sl=slice(None)):
    eng.stage_encoder_inputs(w["x_idx"][sl], w["i_idx"][sl], w["vel"][sl], w["eps"][sl])
    eng.stage_decoder_inputs(B, hist=w["hist"][sl])
    eng.stage_targets(B, w["x_idx"][sl], w["c_idx"][sl])


def _oracle_top_da(orc, p64, cache, batch):
    """d(loss)/d(pre-activations) of the TOP decoder notes layer, (T,B,G*H): what the engine leaves in dec.notes.<top>.da"""
    cell = orc.cfg["cell"]
    out, Y = cache["out"], np.asarray(batch["Y"], np.float64)
    dl = (vo._cce_grad_logits(out["notes"], Y) * cache["g_notes"][..., None]).transpose(1, 0, 2)
    cp, ip, x_seq, hs, cs, acts = cache["dec_notes"][-1]
    da, _, _, _ = vo.rnn_backward(cell, hs, cs, acts, p64[cp + ".U"], dl @ p64["dec.notes.out.W"].T)
    return da


def _check_grads(g, g_o, tol, names=None):
    bad = []
    for k in (names or g_o):
        n = np.linalg.norm(g_o[k])
        if n < 1e-9:
            if np.linalg.norm(g[k]) > 1e-6:
                bad.append((k, "expected zero", float(np.linalg.norm(g[k]))))
        else:
            e = _rel_l2(g[k], g_o[k])
            if not e < tol:
                bad.append((k, e))
    assert not bad, bad


# ----------------------------------------------------------------------------------------------------------------------
# (a), (b): the benched kernels against the oracle, tensor by tensor
# ----------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
@pytest.mark.parametrize("B,T", [(32, 128), (16, 512)])
def test_resident_pipelined_path_gradients_match_oracle(cell, B, T):
    """H=256 bf16, resident + time-pipelined (what bench.py runs; T=512 is its sequence length): losses, every parameter
    gradient and the top decoder layer's d(pre-activation) sequence against the float64 oracle."""
    spec, params, w, batch = _setup(cell, B, T, 4, 64, 2, seed=100 + T)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, w["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    eng = Engine(spec, max_batch=B, dtype="bf16", seed=0)
    assert eng._pipelined(eng.enc_notes) and eng._pipelined(eng.dec_notes), "time-pipelined stacks are off"
    assert eng.fused_latent and eng.tile16
    eng.set_params(params)
    _stage(eng, w, B)
    eng.forward_backward(B)
    eng.check_pipeline()
    m, g = eng.metrics(B), eng.get_grads()
    for k in m_o:
        if not k.endswith("_acc"):
            assert abs(m[k] - m_o[k]) <= 3e-2 * (1 + abs(m_o[k])), (k, m[k], m_o[k])
    _check_grads(g, g_o, 6e-2)
    # the recurrent kernels specifically (the dominant kernels): a tighter look at dU of every layer
    _check_grads(g, g_o, 5e-2, [k for k in g_o if k.endswith(".U")])
    top = eng.dec_notes[-1].prefix
    da = eng._v(top + ".da", T, B, spec.GH).float().cpu().numpy()
    da_o = _oracle_top_da(orc, p64, cache, batch)
    assert _rel_l2(da, da_o) < 5e-2, _rel_l2(da, da_o)
    # ... also early in the sequence, where 500 steps of backward recurrence lie behind the value
    assert _rel_l2(da[:8], da_o[:8]) < 8e-2, _rel_l2(da[:8], da_o[:8])


def test_gru_one_wave_per_simd_kernels_stay_a_working_partner(monkeypatch):
    """MVAE_GRU_W8=0: the engine on the slot-interleaved GRU kernels of rounds 1-5 (seq_layout TILE16P, 4 producer waves per workgroup)
    - the A/B partner of the two-waves-per-SIMD kernels must keep passing what they pass: losses, every
    parameter gradient and the top layer's d(pre-activation) sequence against the float64 oracle, on the same step the default
    engine is checked on (and the default engine really is on the other layout)."""
    from midi_vae_amd import hiplib as hl
    B, T = 32, 128
    spec, params, w, batch = _setup("GRU", B, T, 4, 64, 2, seed=100 + T)
    e0 = Engine(spec, max_batch=B, dtype="bf16", seed=0)
    assert e0._seq_layout(e0.enc_notes[0]) == hl.TILE16Q and e0._rnn_waves(e0.enc_notes[0]) == 8
    del e0
    monkeypatch.setenv("MVAE_GRU_W8", "0")
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, w["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    eng = Engine(spec, max_batch=B, dtype="bf16", seed=0)
    assert eng._seq_layout(eng.enc_notes[0]) == hl.TILE16P and eng._rnn_waves(eng.enc_notes[0]) == 4
    assert eng._pipelined(eng.enc_notes) and eng._pipelined(eng.dec_notes)
    eng.set_params(params)
    _stage(eng, w, B)
    eng.forward_backward(B)
    eng.check_pipeline()
    m, g = eng.metrics(B), eng.get_grads()
    for k in m_o:
        if not k.endswith("_acc"):
            assert abs(m[k] - m_o[k]) <= 3e-2 * (1 + abs(m_o[k])), (k, m[k], m_o[k])
    _check_grads(g, g_o, 6e-2)
    top = eng.dec_notes[-1].prefix
    da = eng._v(top + ".da", T, B, spec.GH).float().cpu().numpy()
    assert _rel_l2(da, _oracle_top_da(orc, p64, cache, batch)) < 5e-2


# ----------------------------------------------------------------------------------------------------------------------
# (c): BASELINE configs[2] / [3] - 4-style, seq_len 256 x 8 voices (T=2048), z=128, 512 windows per GPU
# ----------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_config2_shape_matches_oracle_small_batch(cell):
    """T=2048, z=128, 4 styles, 8 voices at 8 windows: losses and every gradient against the oracle (64 pipeline chunks
    per stack)."""
    B, T = 8, 2048
    spec, params, w, batch = _setup(cell, B, T, 8, 128, 4, seed=7)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, w["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    eng = Engine(spec, max_batch=B, dtype="bf16", seed=0)
    assert eng._pipelined(eng.enc_notes) and eng._pipelined(eng.dec_notes)
    eng.set_params(params)
    _stage(eng, w, B)
    eng.forward_backward(B)
    eng.check_pipeline()
    m, g = eng.metrics(B), eng.get_grads()
    for k in m_o:
        if not k.endswith("_acc"):
            assert abs(m[k] - m_o[k]) <= 3e-2 * (1 + abs(m_o[k])), (k, m[k], m_o[k])
    _check_grads(g, g_o, 8e-2)


def test_config2_full_size_properties():
    """configs[2] / [3] per-GPU share at FULL size (512 windows x T=2048, z=128, C=4, LSTM bf16; ~41 GB resident, activation
    buffers beyond 2 GiB): shard linearity of loss and gradients (rows 0-15 and the LAST 16 rows among the shards checked
    individually), row independence of z and of the argmax decode, determinism, falling loss, no pipeline time-out."""
    B, T, V, Z, C = 512, 2048, 8, 128, 4
    spec, params, w, _ = _setup("LSTM", B, T, V, Z, C, seed=11, hist_scale=0.1)
    eng = Engine(spec, max_batch=B, dtype="bf16", seed=0)
    assert eng._pipelined(eng.enc_notes) and eng._pipelined(eng.dec_notes)
    eng.set_params(params)
    _stage(eng, w, B)
    eng.forward_backward(B)
    eng.check_pipeline()
    m_full = eng.metrics(B)
    g_full = eng.grads.clone()
    z_full = eng.latent(B).copy()
    idx_full = eng.note_indices(B).copy()
    assert np.isfinite(m_full["loss"]) and abs(m_full["notes_loss"] - np.log(61.0)) < 0.3
    assert idx_full.max() <= 60
    top = eng.dec_notes[-1].prefix
    da_tail = eng._v(top + ".da", T, B, spec.GH)[:, B - 16:].float().cpu().numpy()       # the last row tile, every step
    # determinism of the forward pass at full size
    eng.forward_backward(B)
    assert np.array_equal(eng.note_indices(B), idx_full)
    assert _rel_l2(eng.grads.cpu().numpy(), g_full.cpu().numpy().astype(np.float64)) < 1e-4
    # shards of 16 rows in a small engine holding the same parameters
    small = Engine(spec, max_batch=16, dtype="bf16", seed=0)
    small.set_params(params)
    g_sum = torch.zeros_like(g_full)
    losses = {k: 0.0 for k in m_full}
    n_sh = B // 16
    for s in range(n_sh):
        sl = slice(16 * s, 16 * s + 16)
        _stage(small, w, 16, sl)
        small.forward_backward(16)
        ms = small.metrics(16)
        for k in losses:
            losses[k] += ms[k] / n_sh
        g_sum += small.grads
        if s in (0, n_sh - 1):
            assert np.array_equal(small.note_indices(16), idx_full[sl]), s
            np.testing.assert_allclose(small.latent(16), z_full[sl], rtol=0, atol=1e-5)
        if s == n_sh - 1:
            da_s = small._v(top + ".da", T, 16, spec.GH).float().cpu().numpy()
            # batch-mean losses: a row's gradient in a 16-row batch is B/16 times its gradient in the full batch
            assert _rel_l2(da_tail * (B / 16.0), da_s.astype(np.float64)) < 1e-2
    small.check_pipeline()
    for k in losses:
        assert abs(losses[k] - m_full[k]) <= 1e-4 * (1 + abs(m_full[k])), (k, losses[k], m_full[k])
    gs = (g_sum / n_sh).cpu().numpy().astype(np.float64)
    gf = g_full.cpu().numpy()
    lay = eng.layout
    for name in lay.oracle_names():
        a, b = lay.view(gf, name), lay.view(gs, name)
        if np.linalg.norm(b) > 1e-9:
            assert _rel_l2(a, b) < 2e-2, (name, _rel_l2(a, b))
    del small
    # three optimizer steps
    ls = []
    for _ in range(3):
        eng.train_step(B)
        ls.append(eng.metrics(B)["loss"])
    eng.check_pipeline()
    assert all(np.isfinite(ls)) and ls[-1] < ls[0], ls
    del eng
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------------------------------------
# (d): BASELINE configs[4] - decode only, seq_len 512 x 8 voices (T=4096), z=128, 1024 windows per GPU
# ----------------------------------------------------------------------------------------------------------------------

def _decode_inputs(B, Z, seed):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((B, Z)).astype(np.float32)
    z[:, [0, 1]] = z[:, [1, 0]]                                             # latent swap of the style dims
    hist = np.concatenate([np.zeros((1, Z), np.float32), z[:-1]])          # history = previous window's z'
    return z, hist


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_config4_decode_matches_oracle_f32(cell):
    """T=4096 decode of 16 windows in f32 mode: probabilities within 2e-4 of the oracle's, and the fused argmax equals the
    oracle's argmax wherever the oracle's two largest probabilities are at least 1e-5 apart."""
    B, T, V, Z = 16, 4096, 8, 128
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=4, Le=2, Ld=2)
    params = init_params(spec, 5)
    z, hist = _decode_inputs(B, Z, 3)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    out_o = orc.decode(p64, z.astype(np.float64), hist.astype(np.float64),
                       dict(notes=np.zeros((B, 61)), instr=np.zeros((B, 16)), vel=np.zeros((B,))))
    eng = Engine(spec, max_batch=B, dtype="f32", training=False)
    eng.set_params(params)
    eng.stage_decoder_inputs(B, hist=hist, z=z)
    eng.decode(B, want_probs=True)
    out = eng.outputs(B)
    idx = eng.note_indices(B)
    np.testing.assert_allclose(out["notes"], out_o["notes"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out["vel"], out_o["vel"], rtol=2e-4, atol=2e-5)
    assert np.array_equal(idx, np.argmax(out["notes"], -1).astype(np.uint8))        # first max of what the kernel returned
    want = np.argmax(out_o["notes"], -1)
    srt = np.sort(out_o["notes"], -1)
    gap = srt[..., -1] - srt[..., -2]
    differ = idx != want
    # every row whose two largest oracle probabilities are at least 1e-5 apart decodes to the oracle's index; rows inside
    # that band are near-ties of the random initialisation (the decoder is an autonomous system on a constant input, F9: a
    # window can sit on a near-tie for hundreds of steps - measured 1.2 % of these rows), where float32 may pick either
    assert np.all(gap[differ] < 1e-5), (int(differ.sum()), float(gap[differ].max()) if differ.any() else 0.0)
    assert differ.mean() <= np.mean(gap < 1e-5) and differ.mean() < 0.05, (differ.mean(), np.mean(gap < 1e-5))
    # want_probs=False (the configs[4] path: nothing but one byte per row leaves the chip) gives the same indices
    eng.decode(B, want_probs=False)
    assert np.array_equal(eng.note_indices(B), idx)


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_config4_decode_bf16_agreement_with_the_oracle_is_reported(cell):
    """how many of the bf16 path's note indices at T=4096 equal the float64 oracle's argmax?  The decoder is an
    autonomous system on a constant input (F9): bf16 rounding of h feeds back for 4096 steps, so this is a measured agreement,
    not a bit-exact claim (that one is: the index equals the first maximum of the probabilities THIS kernel produced).  Reported
    (printed with -s; committed under profiles/) by the oracle's top-2 gap; asserted: a row decodes differently only where that gap
    is within twice the largest probability error of the bf16 path, and the mean |p - p_oracle| stays below 2e-3."""
    B, T, V, Z = 16, 4096, 8, 128
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=4, Le=2, Ld=2)
    params = init_params(spec, 5)
    z, hist = _decode_inputs(B, Z, 3)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    out_o = orc.decode(p64, z.astype(np.float64), hist.astype(np.float64),
                       dict(notes=np.zeros((B, 61)), instr=np.zeros((B, 16)), vel=np.zeros((B,))))
    eng = Engine(spec, max_batch=B, dtype="bf16", training=False)
    eng.set_params(params)
    eng.stage_decoder_inputs(B, hist=hist, z=z)
    eng.decode(B, want_probs=True)
    eng.check_pipeline()
    probs, idx = eng.outputs(B)["notes"], eng.note_indices(B)
    assert np.array_equal(idx, np.argmax(probs, -1).astype(np.uint8))
    want = np.argmax(out_o["notes"], -1)
    srt = np.sort(out_o["notes"], -1)
    gap = srt[..., -1] - srt[..., -2]
    differ = idx != want
    err_max, err_mean = float(np.max(np.abs(probs - out_o["notes"]))), float(np.mean(np.abs(probs - out_o["notes"])))
    rep = ", ".join("gap >= %g: %.2f %% of %.1f %% rows" % (g, 100 * np.mean(~differ[gap >= g]) if np.any(gap >= g) else float("nan"),
                                                          100 * np.mean(gap >= g)) for g in (0.0, 1e-4, 1e-3, 1e-2))
    print("config4 decode %s bf16 vs float64 oracle, %d rows: argmax agreement by the oracle's top-2 gap: %s; |p - p_oracle| mean %.2e "
          "max %.2e" % (cell, idx.size, rep, err_mean, err_max))
    # a row may decode differently only where the oracle's two largest probabilities are closer than the bf16 path's error band
    assert np.all(gap[differ] <= 2.0 * err_max), (float(gap[differ].max()), err_max)
    assert err_mean < 2e-3, err_mean


def _decisive(params, bias_std=0.5, out_gain=16.0, seed=5):
    """Untrained, the decoder relaxes to a uniform softmax (top-2 gap ~1e-8) and an argmax comparison tests rounding noise
    (VERDICT r04 weak #1): non-zero biases and a larger output kernel make every row DECISIVE, as a trained model's are, while the
    recurrent kernels stay orthogonal - a contractive decoder (tests/studies/decode_rounding.py: with recurrent kernels scaled x5
    the autonomous system is chaotic and not even float32 arithmetic reproduces the float64 indices)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in params.items():
        if k.endswith(".b"):
            v = (rng.standard_normal(v.shape) * bias_std).astype(np.float32)
        elif k.endswith(".out.W"):
            v = (v * out_gain).astype(np.float32)
        out[k] = v
    return out


def _decisive_decode_case(cell, dtype, T, V, capsys, wseed=5, out_gain=16.0, zseed=3):
    B, Z = 16, 128
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=4, Le=2, Ld=2)
    params = _decisive(init_params(spec, wseed), out_gain=out_gain, seed=wseed)
    z, hist = _decode_inputs(B, Z, zseed)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    out_o = orc.decode(p64, z.astype(np.float64), hist.astype(np.float64),
                       dict(notes=np.zeros((B, 61)), instr=np.zeros((B, 16)), vel=np.zeros((B,))))
    want = np.argmax(out_o["notes"], -1)
    srt = np.sort(out_o["notes"], -1)
    gap = srt[..., -1] - srt[..., -2]
    assert np.median(gap) > 1e-3, "the oracle's outputs are not decisive: the comparison would test rounding noise"
    assert len(np.unique(want)) >= 4
    eng = Engine(spec, max_batch=B, dtype=dtype, training=False)
    eng.set_params(params)
    eng.stage_decoder_inputs(B, hist=hist, z=z)
    eng.decode(B, want_probs=True)
    eng.check_pipeline()
    probs, idx = eng.outputs(B)["notes"], eng.note_indices(B)
    assert np.array_equal(idx, np.argmax(probs, -1).astype(np.uint8))          # the index is the first maximum of what the kernel produced
    differ = idx != want
    err = np.abs(probs - out_o["notes"])
    rep = ", ".join("gap >= %g: %.3f %% of %.1f %% rows" % (g, 100 * np.mean(~differ[gap >= g]) if np.any(gap >= g) else float("nan"),
                                                           100 * np.mean(gap >= g)) for g in (0.0, 1e-4, 1e-3, 1e-2, 1e-1))
    with capsys.disabled():
        print("\ndecisive decode %s %s T=%d weights %d gain %g, %d rows: median oracle top-2 gap %.3f; argmax agreement by gap: %s; |p - p_oracle| mean %.2e "
              "max %.2e; rows that differ: %d (largest gap among them %.2e)"
              % (cell, dtype, T, wseed, out_gain, idx.size, np.median(gap), rep, err.mean(), err.max(), int(differ.sum()),
                 float(gap[differ].max()) if differ.any() else 0.0))
    if dtype == "f32":
        assert np.all(gap[differ] < 1e-5), (int(differ.sum()), float(gap[differ].max()))
        assert err.max() < 2e-4
    else:
        assert np.all(gap[differ] <= 2.0 * err.max()), (float(gap[differ].max()), float(err.max()))
        assert not differ[gap >= 0.05].any()
        assert err.mean() < 2e-3, err.mean()
        if out_gain >= 16:
            assert differ.mean() < 5e-3, differ.mean()
    eng.decode(B, want_probs=False)             # (the configs[4] path: one byte per row leaves the chip)
    assert np.array_equal(eng.note_indices(B), idx)
    del eng
    torch.cuda.empty_cache()


@pytest.mark.parametrize("T,V", [(512, 4), (4096, 8)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_decode_on_decisive_weights_equals_the_oracles_argmax(cell, dtype, T, V, capsys):
    """north_star 'bit-exact for the argmax note-index decode', measured where it means something: decisive outputs (median top-2
    gap of the oracle > 1e-3 asserted, 0.2-0.4 measured), T=512 (configs[1]) and T=4096 (configs[4]), the f32 parity mode AND the
    benched bf16 path (reference vae_definition.py:1048-1095 'argmax' decode of decoder.predict).
    f32 mode: every row whose oracle gap is >= 1e-5 decodes to the oracle's index.
    bf16 mode: a row may differ only where the oracle's top-2 gap is within twice the largest probability error of the path - the
    bound the CPU study derives (weights, h and x*W+b rounded to bf16 each move p by ~1e-3..1e-2, none dominates)
    - and every row with a gap >= 0.05 agrees.  The agreement by gap bucket is printed (profiles/r05_*_decode_agreement.txt)."""
    _decisive_decode_case(cell, dtype, T, V, capsys)


@pytest.mark.parametrize("out_gain", [4.0, 16.0])
@pytest.mark.parametrize("wseed,zseed", [(1, 11), (2, 12), (3, 13)])
@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_decisive_decode_over_weight_and_input_seeds(cell, wseed, zseed, out_gain, capsys):
    """the decisive-weights agreement is not one draw - three weight / bias seeds with their own latent inputs and two
    output gains (x4: smaller top-2 gaps, more rows near a tie; x16: the original) at T=512 on the benched bf16 path.  The guarantee
    asserted is the documented one: a row may differ from the float64 oracle only where the oracle's top-2 gap is within twice the
    path's largest probability error, and every row with a gap >= 0.05 agrees."""
    _decisive_decode_case(cell, "bf16", 512, 4, capsys, wseed=wseed, out_gain=out_gain, zseed=zseed)


def test_config4_decode_full_size_properties():
    """configs[4] per-GPU share at full size (1024 windows x T=4096, z=128, LSTM bf16, decode only): deterministic, indices in
    range, and the first / last 16 rows equal the same rows decoded as a 16-row batch (row independence)."""
    B, T, V, Z = 1024, 4096, 8, 128
    spec = ModelSpec(cell="LSTM", H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=4, Le=2, Ld=2)
    params = init_params(spec, 5)
    z, hist = _decode_inputs(B, Z, 4)
    eng = Engine(spec, max_batch=B, dtype="bf16", training=False)
    if T // eng.pipe_chunk > 64:
        eng.pipe_chunk = T // 64
    assert eng._pipelined(eng.dec_notes)
    eng.set_params(params)
    eng.stage_decoder_inputs(B, hist=hist, z=z)
    eng.decode(B, want_probs=False)
    eng.check_pipeline()
    idx = eng.note_indices(B).copy()
    assert idx.shape == (B, T) and idx.max() <= 60
    assert len(np.unique(idx)) > 1
    eng.decode(B, want_probs=False)
    assert np.array_equal(eng.note_indices(B), idx)
    eng.check_pipeline()
    small = Engine(spec, max_batch=16, dtype="bf16", training=False)
    small.pipe_chunk = eng.pipe_chunk
    small.set_params(params)
    for sl in (slice(0, 16), slice(B - 16, B)):
        small.stage_decoder_inputs(16, hist=hist[sl], z=z[sl])
        small.decode(16, want_probs=False)
        # (the initial-state Dense runs as a GEMM whose tile shape depends on the batch: f32 sums in another order can flip a
        # near-tie at random initialisation, hence "nearly all" rather than bit for bit)
        differ = float(np.mean(small.note_indices(16) != idx[sl]))
        assert differ < 0.01, differ
    small.check_pipeline()
    del eng, small
    torch.cuda.empty_cache()
