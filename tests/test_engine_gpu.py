"""End-to-end parity of the device engine against the float64 oracle (GPU box only): losses, every parameter
gradient, parameters after optimizer steps, encoder / decoder outputs, bit-exact argmax decode.

f32 mode  : exact-f32 MFMA + f32 storage.  Tolerance 2e-4 relative (+2e-6 absolute on gradients).
bf16 mode : bf16 MFMA operands and bf16 sequence storage.  Tolerance 3e-2 on losses / outputs; gradients are
            compared by relative L2 error per tensor (< 6e-2).
"""
import numpy as np
import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd import ops
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec, init_params
from oracle.vae_oracle import OracleVAE, make_cfg

pytestmark = pytest.mark.gpu


def _onehot(idx, n):
    out = np.zeros(idx.shape + (n,))
    np.put_along_axis(out, idx[..., None].astype(np.int64), 1, -1)
    return out


def _problem(cell, B, seed=0, H=64, Z=32, T=12, V=4, C=2, **kw):
    spec = ModelSpec(cell=cell, H=H, Z=Z, T=T, V=V, C=C, lr=1e-3, **kw)
    rng = np.random.default_rng(seed)
    params = init_params(spec, seed)
    for k in params:                       # non-zero biases so every path carries signal
        if k.endswith(".b"):
            params[k] = (rng.standard_normal(params[k].shape) * 0.1).astype(np.float32)
    x_idx = np.where(rng.random((B, T)) < 0.35, spec.Dout - 1, rng.integers(0, spec.Dout - 1, (B, T))).astype(np.uint8)
    i_idx = rng.integers(0, spec.ID, (B, V)).astype(np.uint8)
    vel = np.where((x_idx == spec.Dout - 1) | (rng.random((B, T)) < 0.5), 0.0, 0.5 + 0.5 * rng.random((B, T)))
    vel = vel.astype(np.float32)
    hist = (rng.standard_normal((B, Z)) * 0.1).astype(np.float32)
    eps = (rng.standard_normal((B, Z)) * spec.epsilon_std).astype(np.float32)
    c_idx = rng.integers(0, C, (B,)).astype(np.uint8)
    w_notes = np.where(x_idx == spec.Dout - 1, 0.5, 1.0)
    d_idx = (rng.random((B, T)) < 0.4).astype(np.uint8)                   # held-notes roll (reference import_midi.py:267-286)
    n_idx = rng.integers(0, spec.Dout, (B, T)).astype(np.uint8)          # the next window's notes
    sig = (rng.standard_normal((B, spec.SD)) * 0.5).astype(np.float32)        # signature vectors (reference data_class.py:96-221)
    add = rng.standard_normal((B, max(spec.add_dim, 1))).astype(np.float32)[:, :spec.add_dim]
    batch = dict(X=_onehot(x_idx, spec.Din), I=_onehot(i_idx, spec.ID), Vel=vel[..., None].astype(np.float64),
                 Hist=hist.astype(np.float64), Y=_onehot(x_idx, spec.Dout), C=_onehot(c_idx, C), w_notes=w_notes,
                 Held=_onehot(d_idx, 2), Next=_onehot(n_idx, spec.Dout), S=sig.astype(np.float64), Add=add.astype(np.float64))
    raw = dict(x_idx=x_idx, i_idx=i_idx, vel=vel, hist=hist, eps=eps, c_idx=c_idx, w_notes=w_notes, d_idx=d_idx, n_idx=n_idx,
               sig=sig, add=add)
    return spec, params, batch, raw


def _stage(eng, raw, B):
    eng.stage_encoder_inputs(raw["x_idx"], raw["i_idx"], raw["vel"], raw["eps"], d_idx=raw["d_idx"])
    eng.stage_decoder_inputs(B, hist=raw["hist"], add=raw["add"])
    eng.stage_targets(B, raw["x_idx"], raw["c_idx"], w_notes=raw["w_notes"], n_idx=raw["n_idx"], sig=raw["sig"])


def _rel_l2(a, b):
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12)


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "SimpleRNN"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("B", [7, 32])
def test_forward_backward_matches_oracle(cell, dtype, B):
    spec, params, batch, raw = _problem(cell, B, seed=B)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)

    eng = Engine(spec, max_batch=32, dtype=dtype, seed=0)
    eng.set_params(params)
    _stage(eng, raw, B)
    eng.forward_backward(B)
    m = eng.metrics(B)
    g = eng.get_grads()
    tol = 2e-4 if dtype == "f32" else 3e-2
    for k in m_o:
        if k.endswith("_acc"):
            continue
        assert abs(m[k] - m_o[k]) <= tol * (1 + abs(m_o[k])), (k, m[k], m_o[k])
    if dtype == "f32":
        for k in m_o:
            if k.endswith("_acc"):
                assert abs(m[k] - m_o[k]) < 1e-9, (k, m[k], m_o[k])
    for k in g_o:
        if dtype == "f32":
            err = np.abs(g[k] - g_o[k])
            assert np.all(err <= 2e-6 + 2e-4 * np.abs(g_o[k]) + 2e-4 * np.abs(g_o[k]).max()), (k, err.max())
        else:
            if np.linalg.norm(g_o[k]) < 1e-9:
                assert np.linalg.norm(g[k]) < 1e-6, k
            else:
                assert _rel_l2(g[k], g_o[k]) < 6e-2, (k, _rel_l2(g[k], g_o[k]))


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_three_train_steps_match_oracle_f32(cell):
    """ELBO trajectory: same init / data / epsilon on both sides, equal steps (BASELINE target: |dELBO| <= 1e-3)."""
    B = 16
    spec, params, batch, raw = _problem(cell, B, seed=3)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    st = orc.new_opt_state(p64)
    eng = Engine(spec, max_batch=B, dtype="f32")
    eng.set_params(params)
    _stage(eng, raw, B)
    for step in range(3):
        m_o = orc.train_step(p64, st, batch, raw["eps"].astype(np.float64))
        eng.train_step(B)
        m = eng.metrics(B)
        assert abs(m["loss"] - m_o["loss"]) < 1e-3, (step, m["loss"], m_o["loss"])
    got = eng.get_params()
    for k in p64:
        assert np.allclose(got[k], p64[k], rtol=1e-3, atol=2e-5), k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_encode_decode_and_argmax(dtype):
    B = 9
    spec, params, batch, raw = _problem("GRU", B, seed=11)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    eng = Engine(spec, max_batch=16, dtype=dtype, training=False)
    eng.set_params(params)
    eng.stage_encoder_inputs(raw["x_idx"], raw["i_idx"], raw["vel"], raw["eps"])
    z = eng.encode(B).cpu().numpy()
    z_o = orc.encode(p64, batch["X"], batch["I"], batch["Vel"], raw["eps"].astype(np.float64))
    tol = 2e-4 if dtype == "f32" else 3e-2
    assert np.allclose(z, z_o, rtol=tol, atol=tol)
    # decoder alone, latent swap style: feed an arbitrary z
    z_in = z_o.astype(np.float32)
    z_in[:, [0, 1]] = z_in[:, [1, 0]]
    eng.stage_decoder_inputs(B, hist=raw["hist"], z=z_in)
    eng.decode(B, want_probs=True)
    out = eng.outputs(B)
    out_o = orc.decode(p64, z_in.astype(np.float64), raw["hist"].astype(np.float64),
                       dict(notes=np.zeros((B, spec.Dout)), instr=np.zeros((B, spec.ID)), vel=np.zeros((B,))))
    for k in ("notes", "instr", "vel"):
        assert np.allclose(out[k], out_o[k], rtol=tol, atol=tol), k
    # the fused argmax is bit-exact w.r.t. NumPy argmax of the probabilities the engine returned
    assert np.array_equal(eng.note_indices(B), np.argmax(out["notes"], -1).astype(np.uint8))
    if dtype == "f32":
        assert np.array_equal(eng.note_indices(B), np.argmax(out_o["notes"], -1).astype(np.uint8))


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_time_chunk_pipelining_is_exact(cell, dtype):
    """T=64 -> stacked layers run as 4 chunks of 16 steps on separate streams with f32 state carried between
    launches: gradients must match the oracle, and match the un-chunked run of the same engine closely."""
    B = 16
    spec, params, batch, raw = _problem(cell, B, seed=21, T=64)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    res = {}
    for chunks in (4, 1):
        eng = Engine(spec, max_batch=B, dtype=dtype)
        eng.time_chunks = chunks
        assert eng._nchunks(eng.enc_notes) == chunks
        eng.set_params(params)
        _stage(eng, raw, B)
        eng.forward_backward(B)
        res[chunks] = (eng.metrics(B), eng.get_grads())
    m, g = res[4]
    tol = 3e-4 if dtype == "f32" else 3e-2
    assert abs(m["loss"] - m_o["loss"]) <= tol * (1 + abs(m_o["loss"]))
    for k in g_o:
        if np.linalg.norm(g_o[k]) > 1e-9:
            assert _rel_l2(g[k], g_o[k]) < (2e-3 if dtype == "f32" else 8e-2), (k, _rel_l2(g[k], g_o[k]))
        if dtype == "f32" and np.linalg.norm(res[1][1][k]) > 1e-9:
            assert _rel_l2(g[k], res[1][1][k]) < 1e-4, k       # chunked == un-chunked up to atomic-add ordering


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_time_pipelined_stacks_match_chunked_launches(cell):
    """H=256 bf16: the stacked layers run as ONE launch per layer with device-side hand-over every pipe_chunk steps
    (counters polled / published by the running kernels).  Same kernels, same arithmetic as the chunk-per-launch schedule: the
    losses and every gradient must agree up to the atomic-add ordering of the gradient GEMMs, for several steps in a row."""
    B = 32
    spec, params, batch, raw = _problem(cell, B, seed=31, H=256, Z=64, T=64)
    res = {}
    for pipe in (True, False):
        eng = Engine(spec, max_batch=B, dtype="bf16")
        eng.pipeline, eng.pipe_chunk = pipe, 16
        assert eng._pipelined(eng.enc_notes) == pipe
        eng.set_params(params)
        _stage(eng, raw, B)
        out = []
        for _ in range(3):
            eng.forward_backward(B)
            out.append((eng.metrics(B), eng.get_grads()))
        res[pipe] = out
    for (m1, g1), (m0, g0) in zip(res[True], res[False]):
        assert abs(m1["loss"] - m0["loss"]) <= 1e-5 * (1 + abs(m0["loss"]))
        for k in g0:
            if np.linalg.norm(g0[k]) > 1e-9:
                assert _rel_l2(g1[k], g0[k]) < 1e-4, (k, _rel_l2(g1[k], g0[k]))
    # and against the oracle, like every other configuration
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    assert abs(res[True][0][0]["loss"] - m_o["loss"]) <= 3e-2 * (1 + abs(m_o["loss"]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
@pytest.mark.parametrize("B,T,chunk", [(32, 64, 8), (64, 128, 16), (256, 64, 32)])
def test_pipelined_hand_over_never_serves_the_previous_step(cell, B, T, chunk):
    """The hand-over buffers (saved h rows, projected inputs, gate gradients, dX) are REWRITTEN every step at the same addresses,
    and the producers store them write-through with no L2 write-back before the counter (csrc/common.h store16_wt): a consumer
    that hit a stale cache line would read the PREVIOUS step's values.  Two unrelated batches with unrelated weights alternate
    on one engine (sizes that stay L2-resident between steps included); every step must equal the same step of the
    chunk-per-launch schedule, which has a kernel boundary at every hand-over."""
    probs = [_problem(cell, B, seed=s, H=256, Z=64, T=T) for s in (41, 42)]
    res = {}
    for pipe in (True, False):
        eng = Engine(probs[0][0], max_batch=B, dtype="bf16")
        eng.pipeline, eng.pipe_chunk = pipe, chunk
        assert eng._pipelined(eng.enc_notes) == pipe and eng._pipelined(eng.dec_notes) == pipe
        out = []
        for it in range(6):
            spec, params, batch, raw = probs[it & 1]
            eng.set_params(params)
            _stage(eng, raw, B)
            eng.forward_backward(B)
            out.append((eng.metrics(B), eng.get_grads()))
        eng.check_pipeline()
        res[pipe] = out
    for it, ((m1, g1), (m0, g0)) in enumerate(zip(res[True], res[False])):
        assert abs(m1["loss"] - m0["loss"]) <= 1e-5 * (1 + abs(m0["loss"])), it
        for k in g0:
            if np.linalg.norm(g0[k]) > 1e-9:
                assert _rel_l2(g1[k], g0[k]) < 1e-4, (it, k, _rel_l2(g1[k], g0[k]))
    # the two batches really differ (a stale read could not hide)
    assert abs(res[False][0][0]["loss"] - res[False][1][0]["loss"]) > 1e-3


@pytest.mark.parametrize("cell,kw", [("LSTM", {}), ("GRU", dict(H=64, Z=8, T=8)), ("LSTM", dict(H=64, Z=16, T=8))])
def test_fused_latent_chain_matches_separate_launches(cell, kw):
    """The Dense chain around the latent as one launch each way (csrc/latent.hip, f32 FMAs) against the same chain as
    separate GEMM / elementwise launches: losses, z and every gradient agree to f32 round-off; ragged batch included
    (padding rows carry no gradient)."""
    for B in (16, 5):
        spec, params, batch, raw = _problem(cell, B, seed=17, **kw)
        res = {}
        for fused in (True, False):
            eng = Engine(spec, max_batch=16, dtype="f32")
            eng.fused_latent = fused
            eng.set_params(params)
            _stage(eng, raw, B)
            eng.forward_backward(B)
            res[fused] = (eng.metrics(B), eng.get_grads(), eng.latent(B).copy())
            if fused:       # the library took the shape (no silent fallback to the separate launches)
                assert eng._latent_chain_backward(B, eng.pad16(B)) is not None
        (m1, g1, z1), (m0, g0, z0) = res[True], res[False]
        np.testing.assert_allclose(z1, z0, rtol=2e-5, atol=2e-6)
        for k in ("loss", "kl"):
            assert abs(m1[k] - m0[k]) <= 2e-5 * (1 + abs(m0[k])), (k, m1[k], m0[k])
        for k in g0:
            if np.linalg.norm(g0[k]) > 1e-9:
                assert _rel_l2(g1[k], g0[k]) < 2e-4, (B, k, _rel_l2(g1[k], g0[k]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_inference_engine_matches_training_engine_h256_bf16(cell):
    """Engine(training=False) - no saved activations, the kernels' inference variants - on the resident-weights H=256 bf16
    path: evaluate metrics and the argmax decode must equal those of a training engine's forward pass bit for bit (same
    kernels, same arithmetic, only the stores differ), and decoder.predict-style decode on the sampled z likewise."""
    B = 32
    spec, params, batch, raw = _problem(cell, B, seed=23, H=256, Z=64, T=64)
    out = {}
    for training in (True, False):
        eng = Engine(spec, max_batch=B, dtype="bf16", training=training)
        eng.set_params(params)
        _stage(eng, raw, B)
        eng.eval_step(B)
        m, idx = eng.metrics(B), eng.note_indices(B).copy()
        z = eng.latent(B).copy()
        eng.stage_decoder_inputs(B, hist=raw["hist"], z=z)
        eng.decode(B, want_probs=False)
        out[training] = (m, idx, eng.note_indices(B).copy())
    (m1, i1, d1), (m0, i0, d0) = out[True], out[False]
    for k in m1:
        assert m1[k] == pytest.approx(m0[k], rel=1e-6, abs=1e-7), k
    assert np.array_equal(i1, i0) and np.array_equal(d1, d0)
    assert np.array_equal(i0, d0)              # decode on the same z reproduces the autoencoder's notes


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
@pytest.mark.parametrize("slices", [2, 4])
def test_decode_head_following_the_top_layer_slice_by_slice(cell, slices):
    """Engine._head_forward, inference on the per-queue pipelined schedule: the top layer of the notes stack publishes its chunks
    and the argmax head runs slice by slice on the idle gradient queue, each slice released by the chunk that completes it.
    Same kernel on the same rows: the indices equal those of the single head launch behind the stack bit for bit - on a ragged
    batch, twice on one engine (the counters are cumulative), and alternating with the unsliced path."""
    B = 40
    spec, params, batch, raw = _problem(cell, B, seed=29, H=256, Z=64, T=64)
    rng = np.random.default_rng(5)
    zs = [rng.standard_normal((B, spec.Z)).astype(np.float32) for _ in range(3)]
    out = {}
    for n in (1, slices):
        eng = Engine(spec, max_batch=B, dtype="bf16", training=False)
        eng.phase_multi = False              # (the per-queue schedule: what calls of more than phase_max_B windows run)
        eng.head_slices = n
        eng.set_params(params)
        res = []
        for i, z in enumerate(zs):
            if n > 1 and i == 1:
                eng.head_slices = 1          # an unsliced call in between must not disturb the counters of the sliced ones
            eng.stage_decoder_inputs(B, hist=raw["hist"], z=z)
            eng.decode(B, want_probs=False)
            eng.check_pipeline()
            res.append(eng.note_indices(B).copy())
            eng.head_slices = n
        out[n] = res
    for a, b in zip(out[1], out[slices]):
        assert np.array_equal(a, b)
    assert not np.array_equal(out[1][0], out[1][2])     # (different latents decode differently: the comparison is not vacuous)


def test_full_size_step_properties():
    """BASELINE configs[1] at full size (T=512, 256 windows, z=64, LSTM bf16) - too large for the oracle in a test, so
    size-independent properties: the forward pass is deterministic (two evaluations give the same argmax decode bit for bit,
    the same losses up to the order of their atomic sums), the time-pipelined
    schedule equals the chunk-per-launch one, decode(z of the encoder) reproduces the autoencoder's argmax notes (up to
    near-ties), the loss
    is finite, close to ln(61) + small terms at initialisation and falls over three optimizer steps."""
    from midi_vae_amd.synth import make_windows
    B, T = 256, 512
    spec = ModelSpec(cell="LSTM", H=256, Z=64, Din=61, Dout=61, T=T, V=4, ID=16, C=2, Le=2, Ld=2)
    w = make_windows(B, T, 61, 4, 16, 2, 64, seed=77, epsilon_std=spec.epsilon_std)

    def staged(**attrs):
        eng = Engine(spec, max_batch=B, dtype="bf16", seed=5)
        for k, v in attrs.items():
            setattr(eng, k, v)
        eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
        eng.stage_decoder_inputs(B, hist=w["hist"])
        eng.stage_targets(B, w["x_idx"], w["c_idx"])
        return eng

    eng = staged()
    eng.eval_step(B)
    m1, idx1 = eng.metrics(B), eng.note_indices(B).copy()
    eng.eval_step(B)
    m2, idx2 = eng.metrics(B), eng.note_indices(B).copy()
    assert np.array_equal(idx1, idx2)                      # the kernels are deterministic ...
    for k in m1:                                           # ... the loss scalars are sums of atomics: equal to round-off
        assert m1[k] == pytest.approx(m2[k], rel=1e-5, abs=1e-6), (k, m1[k], m2[k])
    assert np.isfinite(m1["loss"]) and abs(m1["notes_loss"] - np.log(61.0)) < 0.2
    z = eng.latent(B).copy()
    eng.stage_decoder_inputs(B, hist=w["hist"], z=z)
    eng.decode(B, want_probs=False)
    # (decode computes the initial-state Denses with a GEMM, the autoencoder pass inside the fused latent chain: the same
    # f32 products summed in another order.  At random initialisation the 61 logits are nearly tied, so a last-bit
    # difference in a state flips the occasional argmax: equal up to a small fraction, not bit for bit.)
    differ = float(np.mean(eng.note_indices(B) != idx1))
    assert differ < 0.02, differ
    chunked = staged(pipeline=False)
    chunked.eval_step(B)
    mc = chunked.metrics(B)
    for k in m1:
        assert m1[k] == pytest.approx(mc[k], rel=1e-5, abs=1e-6), (k, m1[k], mc[k])
    eng.stage_decoder_inputs(B, hist=w["hist"])
    losses = []
    for _ in range(3):
        eng.train_step(B)
        losses.append(eng.metrics(B)["loss"])
    assert all(np.isfinite(losses)) and losses[2] < losses[0], losses
    eng.check_pipeline()              # (raises if a pipelined kernel timed out waiting for its producer)


def test_ragged_batch_reuses_buffers():
    """A smaller last minibatch (songs are not multiples of batch_size) runs in the same engine."""
    spec, params, batch, raw = _problem("GRU", 5, seed=2)
    eng = Engine(spec, max_batch=64, dtype="f32")
    eng.set_params(params)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    for B in (5, 3):
        sub = {k: v[:B] for k, v in raw.items()}
        _stage(eng, sub, B)
        eng.eval_step(B)
        m_o, _ = orc.forward(p64, {k: v[:B] for k, v in batch.items()}, sub["eps"].astype(np.float64))
        assert abs(eng.metrics(B)["loss"] - m_o["loss"]) < 2e-4 * (1 + abs(m_o["loss"]))


def test_h256_one_step_bf16_runs_and_is_finite():
    spec, params, batch, raw = _problem("LSTM", 32, seed=5, H=256, Z=64, T=16)
    eng = Engine(spec, max_batch=32, dtype="bf16")
    eng.set_params(params)
    _stage(eng, raw, 32)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, _ = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    eng.train_step(32)
    m = eng.metrics(32)
    assert np.isfinite(m["loss"]) and abs(m["loss"] - m_o["loss"]) < 3e-2 * (1 + abs(m_o["loss"]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_baseline_config0_elbo_within_1e3_of_oracle(cell):
    """BASELINE configs[0] (seq_len 32 x 4 voices = 128 rows, z=16, 8 windows, H=256, 2+2 layers, every default head) on
    the bf16 resident-weights path: the ELBO of three consecutive optimizer steps within 1e-3 of the float64 oracle - the
    north-star tolerance (tools/config0_parity.py prints the parts and the f32 path as well)."""
    from midi_vae_amd.synth import make_windows
    B, T, V, Z = 8, 128, 4, 16
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=2, Le=2, Ld=2)
    params = init_params(spec, 3)
    w = make_windows(B, T, 61, V, 16, 2, Z, seed=1234, epsilon_std=spec.epsilon_std)
    oh = lambda idx, n: np.eye(n)[idx.astype(np.int64)]
    batch = dict(X=oh(w["x_idx"], 61), I=oh(w["i_idx"], 16), Vel=w["vel"][..., None].astype(np.float64),
                 Hist=w["hist"].astype(np.float64), Y=oh(w["x_idx"], 61), C=oh(w["c_idx"], 2))
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    st = orc.new_opt_state(p64)
    eng = Engine(spec, max_batch=B, dtype="bf16")
    eng.set_params(params)
    eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
    eng.stage_decoder_inputs(B, hist=w["hist"])
    eng.stage_targets(B, w["x_idx"], w["c_idx"])
    for _ in range(3):
        want = orc.train_step(p64, st, batch, w["eps"].astype(np.float64))
        eng.train_step(B)
        got = eng.metrics(B)
        assert abs(got["loss"] - want["loss"]) < 1e-3, (got["loss"], want["loss"])
        assert abs(got["kl"] - want["kl"]) < 1e-4 and abs(got["notes_loss"] - want["notes_loss"]) < 1e-3


@pytest.mark.parametrize("kw", [dict(meta_instrument=False, meta_velocity=False), dict(split=False), dict(extra_layer=False),
                                dict(history=False), dict(C=4), dict(Le=3, Ld=1), dict(Le=1, Ld=3, meta_velocity=False),
                                dict(style=False), dict(extra_layer=False, split=False, meta_instrument=False),
                                dict(meta_held=True, w_held=0.7), dict(meta_next=True, w_next=0.3),
                                dict(meta_held=True, meta_next=True, Ld=1, C=3),
                                dict(meta_held=True, meta_instrument=False, meta_velocity=False),
                                dict(signature=True, SD=5, w_sig=0.6), dict(signature=True, style=False), dict(add_dim=3),
                                dict(comp_notes=True, w_cnotes=0.7), dict(comp_instr=True, w_cinstr=0.4),
                                dict(comp_notes=True, comp_instr=True, signature=True, SD=7, add_dim=2, meta_held=True,
                                     meta_next=True, C=3),
                                dict(bidirectional=True), dict(bidirectional=True, Le=3), dict(bidirectional=True, Le=4, Ld=1)])
@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_model_switches_match_oracle(cell, kw):
    """The settings.py switches that change the graph around the latent and the stacks (no meta encoders / decoders and so no
    pack Dense, un-split or missing extra Dense, no history input, 4 styles, no style head, 1- and 3-layer stacks; the held-notes
    roll + head and the next-notes head of reference vae_definition.py:476-480,648-726, alone - where the reference's pack-Dense
    condition does not fire and the extra Dense takes the 2H concatenation - and together; the signature head, the decoder's
    additional input and the style classifiers on the decoder's notes / instrument OUTPUTS of :737-761, whose gradient re-enters
    the decoder through the softmax of the head they read; the bidirectional encoder of :445-453 - Le-2 Bidirectional(concat)
    layers and one plain layer on top, as written, so Le=2 builds a single plain layer): losses and every gradient of one forward + backward pass against the
    oracle in f32 - also the fused latent chain's variants."""
    B = 8
    spec, params, batch, raw = _problem(cell, B, seed=3, H=64, Z=16, T=8, **kw)
    if not spec.history:
        batch = {k: v for k, v in batch.items() if k != "Hist"}
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    eng = Engine(spec, max_batch=B, dtype="f32", seed=0)
    eng.set_params(params)
    eng.stage_encoder_inputs(raw["x_idx"], raw["i_idx"], raw["vel"], raw["eps"], d_idx=raw["d_idx"])
    eng.stage_decoder_inputs(B, hist=raw["hist"] if spec.history else None, add=raw["add"])
    eng.stage_targets(B, raw["x_idx"], raw["c_idx"], w_notes=raw["w_notes"], n_idx=raw["n_idx"], sig=raw["sig"])
    eng.forward_backward(B)
    m, g = eng.metrics(B), eng.get_grads()
    for k in m_o:
        if not k.endswith("_acc"):
            assert abs(m[k] - m_o[k]) <= 2e-4 * (1 + abs(m_o[k])), (k, m[k], m_o[k])
    for k in g_o:
        err = np.abs(g[k] - g_o[k])
        assert np.all(err <= 2e-6 + 2e-4 * np.abs(g_o[k]) + 2e-4 * np.abs(g_o[k]).max()), (k, err.max())


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_three_layer_stacks_pipelined_h256_bf16(cell):
    """3 + 3 stacked layers on the resident bf16 path: two hand-over interfaces per time-pipelined stack.  Loss against the
    oracle, and against the chunk-per-launch schedule of the same kernels."""
    B = 16
    spec, params, batch, raw = _problem(cell, B, seed=13, H=256, Z=32, T=64, Le=3, Ld=3)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    m_o, _ = orc.forward({k: v.astype(np.float64) for k, v in params.items()}, batch, raw["eps"].astype(np.float64))
    out = {}
    for pipe in (True, False):
        eng = Engine(spec, max_batch=B, dtype="bf16")
        eng.pipeline, eng.pipe_chunk = pipe, 16
        assert eng._pipelined(eng.enc_notes) == pipe and eng._pipelined(eng.dec_notes) == pipe
        eng.set_params(params)
        _stage(eng, raw, B)
        eng.forward_backward(B)
        eng.check_pipeline()
        out[pipe] = (eng.metrics(B), eng.get_grads())
    assert abs(out[True][0]["loss"] - m_o["loss"]) <= 3e-2 * (1 + abs(m_o["loss"]))
    assert abs(out[True][0]["loss"] - out[False][0]["loss"]) <= 1e-5 * (1 + abs(m_o["loss"]))
    for k, g0 in out[False][1].items():
        if np.linalg.norm(g0) > 1e-9:
            assert _rel_l2(out[True][1][k], g0) < 1e-4, (k, _rel_l2(out[True][1][k], g0))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
@pytest.mark.parametrize("bi", [False, True])
def test_optional_heads_on_the_resident_bf16_path(cell, bi):
    """H=256 bf16 with every optional head on (held / next notes, signature, additional decoder input, classifiers on the notes
    and instrument outputs), with a plain and with a bidirectional (Le=3) encoder: loss parts and every gradient against the
    oracle at the bf16 tolerances."""
    B = 16
    spec, params, batch, raw = _problem(cell, B, seed=19, H=256, Z=32, T=32, meta_held=True, meta_next=True, signature=True, SD=6,
                                        add_dim=4, comp_notes=True, comp_instr=True, bidirectional=bi, Le=3 if bi else 2)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    eng = Engine(spec, max_batch=B, dtype="bf16")
    eng.set_params(params)
    _stage(eng, raw, B)
    eng.forward_backward(B)
    eng.check_pipeline()
    m, g = eng.metrics(B), eng.get_grads()
    for k in m_o:
        if not k.endswith("_acc"):
            assert abs(m[k] - m_o[k]) <= 3e-2 * (1 + abs(m_o[k])), (k, m[k], m_o[k])
    for k in g_o:
        if np.linalg.norm(g_o[k]) > 1e-9:
            assert _rel_l2(g[k], g_o[k]) < 8e-2, (k, _rel_l2(g[k], g_o[k]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_batches_on_both_sides_of_the_kstream_gate_on_one_engine(cell):
    """the K-streaming decision depends on the call's batch (residency), the progress counters of the
    pipelined stacks are cumulative over calls.  512 windows (ordinary gradient GEMMs), then a ragged 96 (K-streaming), then 512
    again on ONE engine: every call must finish (a counter row that only some calls advance would leave the next waiter
    waiting for good), leave the status word clean and give the losses of the chunk-per-launch schedule."""
    T, H = 64, 256
    spec, params, batch, raw = _problem(cell, 512, seed=23, H=H, Z=64, T=T)
    eng = Engine(spec, max_batch=512, dtype="bf16")
    ref = Engine(spec, max_batch=512, dtype="bf16")
    ref.pipeline = False
    got, want, ks = [], [], []
    for e, out in ((eng, got), (ref, want)):
        e.set_params(params)
        for B in (512, 96, 512, 96, 96, 512):
            sub = {k: v[:B] for k, v in raw.items()}
            _stage(e, sub, B)
            if e is eng:
                e._cur_B, e._n_side = e.pad16(B), len(e.enc_meta)
                ks.append(e._kstream_ok(e.enc_notes, e.pad16(B)))
            e.train_step(B)
            out.append(e.metrics(B)["loss"])
        e.check_pipeline()
    assert ks == [False, True, False, True, True, False], ks
    for a, b in zip(got, want):
        assert abs(a - b) <= 2e-3 * (1 + abs(b)), (got, want)


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_kstream_partition_rules_accumulate_the_same_gradients_at_full_batch(cell, monkeypatch):
    """Round 6 (Engine.kstream_rows): at 256 windows a published chunk is 4096 k rows and every K-streaming GEMM splits it so that a
    workgroup gets 2048 of them (the 12-tile dW of a dense GRU layer: 24 workgroups) instead of kstream_wgs workgroups per GEMM.  The
    partition only changes who adds what to the f32 accumulators: the gradients of one forward_backward must equal - to accumulation
    order - those of the round-2 rule and of the ordinary gradient GEMMs (K-streaming off), on the same engine inputs."""
    from midi_vae_amd import ops
    B, T = 256, 256                      # (T * B above Engine.defer_grads_rows: the gradient GEMMs run beside the recurrences)
    spec, params, batch, raw = _problem(cell, B, seed=41, H=256, Z=64, T=T)
    launches, real = [], ops.gemm_kstream_multi
    monkeypatch.setattr(ops, "gemm_kstream_multi", lambda problems: (launches.append([max(1, g.split_k) for g in problems]), real(problems))[1])
    grads, parts = {}, {}
    for name, knobs in (("rows", dict(kstream_rows=2048)), ("wgs", dict(kstream_rows=0)), ("off", dict(kstream_grads=False))):
        eng = Engine(spec, max_batch=B, dtype="bf16", seed=0)
        for k, v in knobs.items():
            setattr(eng, k, v)
        eng.set_params(params)
        _stage(eng, raw, B)
        del launches[:]
        eng.forward_backward(B)
        eng.check_pipeline()
        grads[name], parts[name] = eng.get_grads(), [list(p) for p in launches]
        del eng
    assert len(parts["rows"]) == 2 and len(parts["wgs"]) == 2 and parts["off"] == [], parts      # (decoder stack, encoder stack)
    assert all(p == 2 for launch in parts["rows"] for p in launch), parts["rows"]                  # 4096-row chunks in halves
    if cell == "GRU":
        assert parts["wgs"] != parts["rows"], parts                                                # (the round-2 rule: 1, 2 and 4)
    for other in ("wgs", "off"):
        for k, g in grads["rows"].items():
            n = np.linalg.norm(grads[other][k])
            if n > 1e-9:
                assert _rel_l2(g, grads[other][k]) < 2e-4, (other, k, _rel_l2(g, grads[other][k]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_collected_gemms_in_front_of_the_joins_change_no_gradient(cell):
    """Round 6 (Engine.flush_before_join): at short sequences the encoder's collected weight-gradient GEMMs go out BEFORE the critical
    queue joins the gradient queues - they read only what the encoder's BPTT launch wrote on that queue.  Same gradients (to f32
    accumulation order) and, over three optimizer steps, the same parameters as with the GEMMs behind the joins."""
    B, T = 256, 64                       # the reference's shipped shape: T * B = 16384 rows, a step that collects its GEMMs
    spec, params, batch, raw = _problem(cell, B, seed=43, H=256, Z=256 if cell == "GRU" else 64, T=T)
    out = {}
    for flag in (True, False):
        eng = Engine(spec, max_batch=B, dtype="bf16", seed=0)
        eng.flush_before_join = flag
        assert eng._defers_grads(B)
        eng.set_params(params)
        _stage(eng, raw, B)
        eng.forward_backward(B)
        eng.check_pipeline()
        g = eng.get_grads()
        for _ in range(3):
            eng.train_step(B)
        eng.check_pipeline()
        out[flag] = (g, eng.get_params(), eng.metrics(B)["loss"])
        del eng
    for k, g in out[True][0].items():
        n = np.linalg.norm(out[False][0][k])
        if n > 1e-9:
            assert _rel_l2(g, out[False][0][k]) < 2e-4, (k, _rel_l2(g, out[False][0][k]))
    # (f32 atomics make the accumulation order - and with Adam's m / sqrt(v) the update of an element whose gradient is rounding noise -
    #  differ from run to run whatever the flag: the three steps' UPDATE is compared, not the parameters element by element)
    for k, v in out[True][1].items():
        upd_t, upd_f = v - params[k], out[False][1][k] - params[k]
        if np.linalg.norm(upd_f) > 1e-9:
            assert _rel_l2(upd_t, upd_f) < 5e-2, (k, _rel_l2(upd_t, upd_f))
    assert abs(out[True][2] - out[False][2]) < 1e-3 * (1 + abs(out[False][2]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_elbo_trajectory_fresh_epsilon_on_the_benched_schedule(cell):
    """north_star: 'ELBO within 1e-3 of reference after equal steps'.  The schedule bench.py times - H=256 bf16, resident
    slot-interleaved kernels, time-pipelined stacks, K-streaming gradient launch, fused latent chain - at its sequence length
    (T=512), 16 windows, TEN optimizer steps with a FRESH epsilon per step and lr 1e-3 (5x the reference's, so the parameters
    move), against the float64 oracle stepping from the same initial parameters on the same draws: the ELBO (Keras total loss)
    and each of its parts within 1e-3 at every step; parameters: see the assertion."""
    B, T, steps = 16, 512, 10
    spec, params, batch, raw = _problem(cell, B, seed=31, H=256, Z=64, T=T, epsilon_std=0.1)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p = {k: v.astype(np.float64) for k, v in params.items()}
    p0 = {k: v.copy() for k, v in p.items()}
    st = orc.new_opt_state(p)
    eng = Engine(spec, max_batch=B, dtype="bf16")
    eng.set_params(params)
    assert eng._pipelined(eng.enc_notes) and eng._pipelined(eng.dec_notes)
    eng._cur_B, eng._n_side = 16, len(eng.enc_meta)
    assert eng._kstream_ok(eng.enc_notes, 16)
    worst = {}
    for i in range(steps):
        eps = (np.random.default_rng(1000 + i).standard_normal((B, spec.Z)) * spec.epsilon_std).astype(np.float32)
        m_o = orc.train_step(p, st, batch, eps.astype(np.float64))
        raw["eps"] = eps
        _stage(eng, raw, B)
        eng.train_step(B)
        m = eng.metrics(B)
        for k in ("loss", "notes_loss", "instr_loss", "vel_loss", "style_loss", "kl"):
            worst[k] = max(worst.get(k, 0.0), abs(m[k] - m_o[k]))
    eng.check_pipeline()
    print("max |engine - oracle| over %d steps: %s" % (steps, {k: "%.2e" % v for k, v in worst.items()}))
    assert all(v <= 1e-3 for v in worst.values()), worst
    # parameters after ten Adam steps: the UPDATE (p - p0) of every tensor against the oracle's.  Adam normalises each element's
    # step to ~lr whatever the size of its gradient, so an element whose gradient is at the bf16 noise floor moves by lr in a
    # noisy direction on both sides; stated tolerance: relative L2 of the update < 0.35 per tensor, and no element further than
    # 2 * steps * lr from the oracle's (the bound if every step went the opposite way).
    got = eng.get_params()
    rel = {}
    for k in p:
        du_o, du = p[k] - p0[k], got[k].astype(np.float64) - p0[k]
        if np.linalg.norm(du_o) > 0:
            rel[k] = float(np.linalg.norm(du - du_o) / np.linalg.norm(du_o))
        assert np.max(np.abs(du - du_o)) <= 2 * steps * spec.lr + 1e-7, k
    print("relative L2 error of the %d-step update, worst tensors:" % steps, sorted(rel.items(), key=lambda kv: -kv[1])[:5])
    assert max(rel.values()) < 0.35, sorted(rel.items(), key=lambda kv: -kv[1])[:5]


def test_elbo_stays_in_the_band_over_forty_steps_at_the_reference_learning_rate():
    """The long run of tests/studies/elbo_long.py (200 steps, profiles/r05_*_elbo_200.txt) at a reduced size: 40 Adam steps at the
    reference's lr 2e-4 (settings.py:113), LSTM bf16 on the benched schedule, T=128 - the ELBO within 1e-3 of the oracle's at EVERY
    step, and no drift: the last ten steps are no further from the oracle than 3x the first ten."""
    from tests.studies.elbo_long import run
    import io
    worst, left, rows = run("LSTM", 2e-4, 40, T=128, out=io.StringIO())
    assert left is None and worst <= 1e-3, (worst, left)
    first, last = max(r[3]["loss"] for r in rows[:10]), max(r[3]["loss"] for r in rows[-10:])
    assert last <= max(3 * first, 2e-4), (first, last)


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
@pytest.mark.parametrize("B", [32, 200])
def test_phase_launches_equal_the_per_stream_schedule(cell, B):
    """engine_phases.py: every recurrence of a phase as ONE launch on the critical queue (the velocity roll's x*W + b expansion a
    producer inside the encoder-forward launch) against the per-stream schedule (one launch per recurrence and queue, events
    between them).  Same kernel bodies, same arithmetic: the saved activations and every sequence the BPTT reads are bit-identical,
    losses and gradients agree to the order of their atomic sums - over three train steps, ragged batch included (B = 200)."""
    import torch
    spec, params, batch, raw = _problem(cell, B, seed=57, H=256, Z=64, T=64)
    res = {}
    for multi in (True, False):
        eng = Engine(spec, max_batch=B, dtype="bf16")
        eng.phase_multi = multi
        eng.set_params(params)
        _stage(eng, raw, B)
        eng.forward_backward(B)
        eng.check_pipeline()
        seqs = {k: eng.store[k].clone() for k in ("enc.vel.xp", "enc.vel.hs", "enc.notes.1.hs", "enc.notes.1.acts", "dec.notes.1.hs",
                                                  "dec.notes.1.acts", "dec.notes.0.da", "enc.notes.0.da", "enc.vel.da", "enc.instr.da")}
        m0, g0 = eng.metrics(B), eng.get_grads()
        losses = []
        for _ in range(3):
            eng.train_step(B)
            losses.append(eng.metrics(B)["loss"])
        eng.check_pipeline()
        res[multi] = (seqs, m0, g0, losses)
    (s1, m1, g1, l1), (s0, m0, g0, l0) = res[True], res[False]
    for k in s0:
        assert torch.equal(s1[k], s0[k]), k
    for k in m0:
        assert m1[k] == pytest.approx(m0[k], rel=1e-5, abs=1e-6), k
    for k in g0:
        assert _rel_l2(g1[k], g0[k]) < 1e-4 or np.linalg.norm(g0[k]) < 1e-9, k
    for a, b in zip(l1, l0):
        assert abs(a - b) <= 1e-4 * (1 + abs(b)), (l1, l0)


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_one_hot_bottom_layer_written_out_equals_the_indexed_kernel(cell, monkeypatch):
    """Engine._index_as_dense: inside the encoder-forward phase launch the table rows of the one-hot bottom layer are written out
    by a chunk-publishing producer (mvae_xpand_args.idx / table) and the layer runs on the dense-input kernel; the default is
    per cell (GRU on, LSTM off - profiles/r03_r_index_dense.txt).  Both settings for both cells: the written-out projection
    equals the gathered table rows bit for bit, so every saved sequence and the whole train step do too (ragged batch)."""
    import torch
    B = 200
    spec, params, batch, raw = _problem(cell, B, seed=61, H=256, Z=64, T=64)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setattr(Engine, "INDEX_DENSE", flag == "1")
        eng = Engine(spec, max_batch=B, dtype="bf16")
        assert eng.index_dense == (flag == "1")
        eng.set_params(params)
        _stage(eng, raw, B)
        eng.forward_backward(B)
        eng.check_pipeline()
        seqs = {k: eng.store[k].clone() for k in ("enc.notes.0.hs", "enc.notes.0.acts", "enc.notes.1.hs", "enc.notes.0.da")}
        if flag == "1":     # the producer's output against the table rows it names
            Bp = eng._cur_B
            xp = eng._v("enc.notes.0.xp", spec.T, Bp, spec.GH)
            idx = eng._v("in.x_idx", spec.T, Bp).long()
            table = eng._v("enc.notes.0.table", eng.enc_notes[0].K, spec.GH)
            R, N = spec.T * Bp, spec.GH      # TILE16: 16x16 tiles, lane = row % 16 + 16 * (col % 16 // 4), 4 columns per lane
            rows = xp.reshape(R // 16, N // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(R, N)
            assert torch.equal(rows, table[idx.view(-1)])
        m, g = eng.metrics(B), eng.get_grads()
        eng.train_step(B)
        res[flag] = (seqs, m, g, eng.metrics(B)["loss"])
    (s1, m1, g1, l1), (s0, m0, g0, l0) = res["1"], res["0"]
    for k in s0:
        assert torch.equal(s1[k], s0[k]), k
    for k in m0:
        assert m1[k] == pytest.approx(m0[k], rel=1e-5, abs=1e-6), k
    for k in g0:
        assert _rel_l2(g1[k], g0[k]) < 1e-4 or np.linalg.norm(g0[k]) < 1e-9, k
    assert abs(l1 - l0) <= 1e-4 * (1 + abs(l0))


@pytest.mark.parametrize("dtype,H", [("f32", 64), ("bf16", 256)])
def test_nonzero_decoder_start_rows_match_oracle(dtype, H):
    """the constant input of the decoder cells (reference vae_definition.py:820,916 always passes zeros): an
    all-zero start reads the bias rows the weight preparation wrote, a NON-zero one takes the start*W + b GEMM and the
    start^T dxp0 weight gradient - both against the oracle, alternating on one engine (zero, non-zero, zero)."""
    B = 16
    spec, params, batch, raw = _problem("LSTM", B, seed=71, H=H, Z=32, T=16)
    rng = np.random.default_rng(5)
    starts = dict(start_notes=rng.standard_normal((B, spec.Dout)).astype(np.float32) * 0.3,
                  start_instr=rng.standard_normal((B, spec.ID)).astype(np.float32) * 0.3,
                  start_vel=rng.standard_normal((B,)).astype(np.float32) * 0.3)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    eng = Engine(spec, max_batch=B, dtype=dtype)
    eng.set_params(params)
    tol_l, tol_g = (2e-4, 2e-3) if dtype == "f32" else (3e-2, 6e-2)
    for use in (False, True, False):
        b2 = dict(batch, **({k: v.astype(np.float64) for k, v in starts.items()} if use else {}))
        m_o, cache = orc.forward(p64, b2, raw["eps"].astype(np.float64))
        g_o = orc.backward(p64, cache)
        _stage(eng, raw, B)
        eng.stage_decoder_inputs(B, hist=raw["hist"], add=raw["add"], **(starts if use else {}))
        assert eng.start_zero["dec.notes.0"] == (not use)
        eng.forward_backward(B)
        m, g = eng.metrics(B), eng.get_grads()
        assert abs(m["loss"] - m_o["loss"]) <= tol_l * (1 + abs(m_o["loss"])), (use, m["loss"], m_o["loss"])
        for k in ("dec.notes.0.W", "dec.notes.0.b", "dec.vel.cell.W", "dec.instr.cell.W", "dec.notes.0.U"):
            n = np.linalg.norm(g_o[k])
            if n < 1e-12:
                assert np.linalg.norm(g[k]) < 1e-6, (use, k)
            else:
                assert _rel_l2(g[k], g_o[k]) < tol_g, (use, k, _rel_l2(g[k], g_o[k]))


def test_twenty_engines_in_one_process_keep_their_queues_apart():
    """The runtime deals streams onto a fixed number of hardware queues; an engine whose projection stream landed on the critical
    stream's queue would stall its phase launches until the time-out (Engine._own_queue_stream asks the runtime by experiment when
    the streams are created).  Twenty engines created and dropped in one process - every one a new set of streams - each run a
    train step on the time-pipelined phase launches: no fallback warning, no time-out status."""
    import gc
    import warnings
    spec, params, batch, raw = _problem("LSTM", 16, seed=3, H=256, Z=64, T=32)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for i in range(20):
            eng = Engine(spec, max_batch=16, dtype="bf16")
            eng.set_params(params)
            _stage(eng, raw, 16)
            eng.train_step(16)
            assert np.isfinite(eng.metrics(16)["loss"])
            eng.check_pipeline()
            assert eng.pipeline
            del eng
            gc.collect()
    assert not [str(w.message) for w in rec if "timed out" in str(w.message)], [str(w.message) for w in rec]


@pytest.mark.parametrize("cell,dtype,H,T", [("GRU", "f32", 64, 12), ("LSTM", "f32", 64, 12), ("LSTM", "bf16", 256, 64)])
def test_attach_instruments_two_hot_rows_match_oracle(cell, dtype, H, T):
    """attach_instruments (reference import_midi.py:288-292, settings.py:186-187,207-208): every notes row carries its voice's
    instrument one-hot behind the pitch one-hot - input_dim = output_dim = 61 + 16, two-hot rows.  Encoder layer 1 then sums two
    rows of its kernel (written out by mvae_gather2_tile16), the notes head's cross-entropy has two target columns
    (-log p[pitch] - log p[instrument], accuracy against the first hot column) and W's gradient two one-hot GEMMs.  Losses and
    every gradient against the oracle, which multiplies the dense two-hot rows; f32: 2e-4 / 2e-3, bf16 resident path: 3e-2 / 6e-2."""
    B, A = 16, 16
    spec, params, batch, raw = _problem(cell, B, seed=91, H=H, Z=32, T=T, Din=77, Dout=77, attach=A)
    D0 = spec.Din - A
    rng = np.random.default_rng(4)
    x_idx = rng.integers(0, D0, (B, T)).astype(np.uint8)
    xa_idx = np.tile(raw["i_idx"][:, np.arange(T) % spec.V], 1).astype(np.uint8)         # row t = voice t % V's instrument category
    X = np.concatenate([_onehot(x_idx, D0), _onehot(xa_idx, A)], -1)
    batch = dict(batch, X=X, Y=X)
    raw = dict(raw, x_idx=x_idx)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    eng = Engine(spec, max_batch=B, dtype=dtype)
    eng.set_params(params)
    eng.stage_encoder_inputs(x_idx, raw["i_idx"], raw["vel"], raw["eps"], d_idx=raw["d_idx"], xa_idx=xa_idx)
    eng.stage_decoder_inputs(B, hist=raw["hist"], add=raw["add"])
    eng.stage_targets(B, x_idx, raw["c_idx"], w_notes=raw["w_notes"], n_idx=raw["n_idx"], sig=raw["sig"], ya_idx=xa_idx)
    eng.forward_backward(B)
    eng.check_pipeline()
    m, g = eng.metrics(B), eng.get_grads()
    tol_l, tol_g = (2e-4, 2e-3) if dtype == "f32" else (3e-2, 6e-2)
    for k in m_o:
        assert abs(m[k] - m_o[k]) <= tol_l * (1 + abs(m_o[k])) + (1e-6 if k.endswith("_acc") else 0), (k, m[k], m_o[k])
    for k in g_o:
        n = np.linalg.norm(g_o[k])
        if n < 1e-12:
            assert np.linalg.norm(g[k]) < 1e-6, k
        else:
            assert _rel_l2(g[k], g_o[k]) < tol_g, (k, _rel_l2(g[k], g_o[k]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_gradient_time_portions_equal_whole_sequence_gradients(cell):
    """Per-queue schedule, long sequences: a layer's parameter-gradient GEMMs are released in portions of the time axis by the da
    chunks its BPTT publishes (Engine._grad_portions; BASELINE configs[2]/[3]: T=2048 x 512 windows).  Forced to 4 portions at a
    small shape: every gradient tensor equals the whole-sequence launch (to f32 summation order), the single-layer branches
    (velocity encoder / decoder) publish and follow the same way, and eight optimizer steps - the last ones plan replays - keep the
    counters in step (a mis-gated portion reads da before it is written: the gradients would show it)."""
    B = 32
    spec, params, _, raw = _problem(cell, B, seed=9, H=256, Z=32, T=128)
    grads = {}
    for portions in (0, 4):
        eng = Engine(spec, max_batch=B, dtype="bf16", seed=0)
        eng.phase_multi, eng.grad_portions = False, portions        # (the schedule batches above 256 windows take)
        eng.set_params(params)
        _stage(eng, raw, B)
        eng.forward_backward(B)
        grads[portions] = eng.get_grads()
        eng.check_pipeline()
        for _ in range(8):          # (steps 1-2 start from other engine states, 3-5 are recorded, 6-8 replayed)
            eng.train_step(B)
        eng.check_pipeline()
        assert np.isfinite(eng.metrics(B)["loss"])
        if portions:
            assert eng.plan_stats["replayed"] >= 1, eng.plan_stats
    for k in grads[0]:
        ref = np.linalg.norm(grads[0][k])
        if ref > 1e-8:
            assert np.linalg.norm(grads[4][k] - grads[0][k]) <= 2e-3 * ref, (k, np.linalg.norm(grads[4][k] - grads[0][k]) / ref)
