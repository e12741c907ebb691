"""Step plans on the device (include/midivae_hip.h 'STEP PLANS'; midi-vae_amd/plan.py, engine_plan.py): a replayed step enqueues
what the Python enqueue would have - same losses, same parameters, counters in step with the kernels that wait for them - and
every kind of call the drop-in makes (fit's two halves, evaluate, encoder.predict, decoder.predict) is replayable.
Reference: ONE backend call per minibatch behind autoencoder.fit / evaluate / predict (vae_training.py:804-809, :300, :289,795)."""
import numpy as np
import pytest
import torch

import midi_vae_amd  # noqa: F401
from midi_vae_amd import plan as P
from midi_vae_amd.engine import Engine
from test_engine_gpu import _problem, _stage

pytestmark = pytest.mark.gpu


def _engine(spec, params, raw, B, plans, dtype="bf16", **kw):
    eng = Engine(spec, max_batch=B, dtype=dtype, seed=0, **kw)
    eng.use_plans = plans
    eng.set_params(params)
    _stage(eng, raw, B)
    return eng


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
@pytest.mark.parametrize("H,T,dtype", [(256, 64, "bf16"), (64, 12, "f32")])
def test_replayed_train_steps_equal_the_python_enqueue(cell, H, T, dtype):
    """8 optimizer steps: enqueued from Python / 3 recorded then 5 replayed.  The resident H=256 path runs its time-pipelined
    phase launches (kernels waiting on counters the plan patches): a mis-advanced counter is a time-out, which check_pipeline
    raises on."""
    B = 32
    spec, params, _, raw = _problem(cell, B, seed=3, H=H, Z=32, T=T)
    losses = {}
    for plans in (False, True):
        eng = _engine(spec, params, raw, B, plans, dtype)
        out = []
        for _ in range(8):
            eng.train_step(B)
            out.append(eng.metrics(B)["loss"])
        eng.check_pipeline()
        losses[plans] = (out, eng.get_params())
        if plans:
            assert eng.plan_stats["refused"] == {} or all("torch operation" in v for v in eng.plan_stats["refused"].values()), eng.plan_stats
            assert eng.plan_stats["replayed"] >= 4, eng.plan_stats
    (la, pa), (lb, pb) = losses[False], losses[True]
    # (the gradient GEMMs reduce with f32 atomics: two runs of the SAME enqueue differ in the last bits, so not bit-identical)
    np.testing.assert_allclose(lb, la, rtol=2e-5, atol=2e-6)
    # parameters: per tensor, the two runs' 8-step updates agree to a few percent in L2 (Keras Adam turns the sign of a gradient
    # element that is rounding noise into a full +-lr step, so single elements may differ by what two runs of the same enqueue do)
    for k in pa:
        upd = np.linalg.norm(pa[k] - params[k])
        assert np.linalg.norm(pa[k] - pb[k]) <= 5e-2 * upd + 1e-6, (k, np.linalg.norm(pa[k] - pb[k]), upd)


def test_replays_and_python_steps_alternate_on_one_engine():
    """the plan's patches are relative to the engine's counters, not to a run index: Python-enqueued steps (here: every step whose
    launches are bracketed for profiling, as bench.py does every 4th step) may come between replays"""
    B = 32
    spec, params, _, raw = _problem("LSTM", B, seed=4, H=256, Z=32, T=64)
    eng = _engine(spec, params, raw, B, True)
    for i in range(12):
        if i >= 4 and i % 3 == 0:
            eng.prof, eng.prof_kinds = {}, {("rnn_bwd_multi", "dec")}
        eng.train_step(B)
        eng.prof = None
    torch.cuda.synchronize()
    eng.check_pipeline()
    assert eng.plan_stats["replayed"] >= 5 and np.isfinite(eng.metrics(B)["loss"])


def test_the_plan_holds_what_python_would_enqueue_next():
    """arm a plan, then let Python enqueue the next step under a Recorder: the calls it makes are the calls the plan holds, the
    fields that differ are exactly the plan's patches evaluated at the current counters"""
    B = 16
    spec, params, _, raw = _problem("GRU", B, seed=5, H=256, Z=32, T=32)
    eng = _engine(spec, params, raw, B, True)
    for _ in range(4):          # (step 1 zeroes the gradient buffer with a torch fill: recorded under another key, refused)
        eng.train_step(B)
    slot = [s for s in eng._plans.values() if s.plan is not None]
    assert len(slot) == 1, eng.plan_stats
    plan = slot[0].plan
    before = eng._plan_counters()
    eng.use_plans = False
    with P.Recorder() as rec:
        eng.train_step(B)
    assert rec.tainted is None
    assert len(rec.calls) == plan.n_calls
    # the same step through the constructor's own comparison: recording it three times over must give the same patches
    before.update(eng._call_params(B))        # (the call's parameters: real window count, 1 / global minibatch - always patches)
    again = P.StepPlan([(rec.calls, rec.tags, before, eng._plan_counters())] * 3, [rec.marks] * 3)
    n_param = sum(1 for k in rec.tags.values() if k[0] == "param")
    assert again.n_patches == n_param > 0 and again.n_calls == plan.n_calls       # (identical recordings: everything else constant)
    again.close()
    eng.check_pipeline()


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_fit_halves_evaluate_encode_decode_are_replayable(cell):
    """the calls model.py makes: train_step_begin / train_step_finish (targets staged between them), eval_step, encode, decode -
    five of each, outputs of the replays equal to the recorded calls' (forward kernels are deterministic: bit for bit)"""
    B = 32
    spec, params, _, raw = _problem(cell, B, seed=6, H=256, Z=32, T=64)
    eng = _engine(spec, params, raw, B, True)
    for _ in range(6):
        eng.train_step_begin(B)
        eng.train_step_finish(B)
    eng.check_pipeline()
    assert eng.plan_stats["replayed"] >= 4, eng.plan_stats
    inf = Engine(spec, max_batch=B, dtype="bf16", seed=0, training=False)
    inf.set_params(params)
    _stage(inf, raw, B)
    outs = {"eval": [], "encode": [], "decode": []}
    for _ in range(6):
        inf.eval_step(B)
        outs["eval"].append(inf.metrics(B)["loss"])
        outs["encode"].append(inf.encode(B).float().cpu().numpy().copy())
        inf.stage_decoder_inputs(B, hist=raw["hist"], z=raw["hist"], add=raw["add"])
        inf.decode(B, want_probs=False)
        outs["decode"].append(inf.note_indices(B).copy())
    inf.check_pipeline()
    assert inf.plan_stats["replayed"] >= 6, inf.plan_stats
    for k, v in outs.items():
        for x in v[1:]:
            if k == "eval":      # (the loss is a sum of f32 atomics over the rows: equal to the last bits, not bit for bit)
                np.testing.assert_allclose(x, v[0], rtol=1e-6, err_msg=k)
            else:
                np.testing.assert_array_equal(np.asarray(x), np.asarray(v[0]), err_msg=k)


def test_reference_shipped_configuration_matches_the_oracle_and_replays():
    """the configuration the reference ships (settings.py:108-112,140,155; models/BvM/params.txt): GRU, T = 16 x 4 = 64, latent
    256, batch 256 - on the resident bf16 path against the float64 oracle: losses, every gradient tensor, and the parameters
    after 3 optimizer steps (the third and later steps of a run are plan replays)."""
    from oracle.vae_oracle import OracleVAE, make_cfg
    B = 256
    spec, params, batch, raw = _problem("GRU", B, seed=7, H=256, Z=256, T=64)
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    m_o, cache = orc.forward(p64, batch, raw["eps"].astype(np.float64))
    g_o = orc.backward(p64, cache)
    eng = _engine(spec, params, raw, B, True)
    assert eng._pipelined(eng.enc_notes), "the resident time-pipelined path is off"
    eng.forward_backward(B)
    m, g = eng.metrics(B), eng.get_grads()
    for k in m_o:
        if not k.endswith("_acc"):
            assert abs(m[k] - m_o[k]) <= 3e-2 * (1 + abs(m_o[k])), (k, m[k], m_o[k])
    for k in g_o:
        if np.linalg.norm(g_o[k]) > 1e-8:
            err = np.linalg.norm(g[k] - g_o[k]) / np.linalg.norm(g_o[k])
            assert err < 6e-2, (k, err)
    # 3 optimizer steps (Keras Adam) against the oracle's: ELBO at every step, the parameter update afterwards; 3 more steps are replays
    eng2 = _engine(spec, params, raw, B, True)
    p, st = {k: v.copy() for k, v in p64.items()}, None
    st = orc.new_opt_state(p)
    for i in range(6):
        eng2.train_step(B)
        if i < 3:
            m_o = orc.train_step(p, st, batch, raw["eps"].astype(np.float64))
            assert abs(eng2.metrics(B)["loss"] - m_o["loss"]) <= 1e-3 * (1 + abs(m_o["loss"])), (i, eng2.metrics(B)["loss"], m_o["loss"])
        if i == 2:
            got = eng2.get_params()
            for k in p:
                d = p[k] - p64[k]
                if np.linalg.norm(d) > 1e-9:
                    assert np.linalg.norm(got[k].astype(np.float64) - p[k]) <= 8e-2 * np.linalg.norm(d) + 1e-7, k
    eng2.check_pipeline()
    assert eng2.plan_stats["replayed"] >= 2, eng2.plan_stats


def test_fused_history_prepass_steps_replay(monkeypatch):
    """the reference's loop on songs of one minibatch (vae_training.py:788-809: encoder.predict, roll, fit) with the history
    pre-pass fused into the train step: z' lands in a fixed engine buffer, so these steps are plans too - same histories and
    weights as the Python enqueue, song after song"""
    from midi_vae_amd import packers as pk
    from midi_vae_amd.config import build_settings, create_kwargs
    from midi_vae_amd.model import VAE
    from midi_vae_amd.synth import make_windows, to_reference_format
    s = build_settings(cell_type="GRU", lstm_size=256, latent_dim=32, input_length=8, output_length=8, batch_size=16,
                       learning_rate=1e-3)
    n = 16
    songs = []
    for i in range(7):
        w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=20 + i)
        songs.append(to_reference_format(w))
    res = {}
    for plans in ("0", "1"):
        monkeypatch.setenv("MVAE_PLANS", plans)
        m = VAE().create(compute_dtype="bf16", seed=2, **create_kwargs(s))
        losses = []
        for (X, Y, C, I, V, D) in songs:
            H = m.encoder.predict(pk.prepare_encoder_input_list(s, X, I, V, D), batch_size=16, device=True)
            x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, np.zeros((n, s["signature_vector_length"])), H,
                                                                    return_sample_weight=True)
            h = m.autoencoder.fit(x, y, epochs=1, batch_size=16, shuffle=False, sample_weight=sw, verbose=False)
            losses.append(h.history["loss"][0])
            assert np.all(np.isfinite(H.latent()))          # z' of the fused pre-pass reached the caller's rows
        eng = m._shared.engine
        eng.check_pipeline()
        res[plans] = (losses, m.autoencoder.get_weights(), dict(eng.plan_stats))
    assert res["0"][2]["replayed"] == 0 and res["1"][2]["replayed"] >= 4, (res["0"][2], res["1"][2])
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=3e-5, atol=3e-6)
    for a, b in zip(res["0"][1], res["1"][1]):
        assert np.linalg.norm(a - b) <= 5e-2 * np.linalg.norm(a) * 1e-2 + 1e-4, np.linalg.norm(a - b)


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_short_sequences_defer_their_weight_gradient_gemms_into_one_launch(cell):
    """Engine.defer_grads_rows (round 5): at the reference's shipped length (T = 64, settings.py:108-109) the weight-gradient GEMMs of
    a step are collected during the backward pass and leave as mvae_gemm_multi launches: the decoder side's on the gradient queue
    BESIDE the encoder BPTT launch (defer_early, released by that launch's first published chunk), the encoder's behind the last
    recurrence - ONE launch behind it with defer_early off.  Same GEMM bodies, same operands: losses identical, every gradient tensor equal to the order of the split-K atomics, against the
    schedule that launches them one by one beside the recurrences (defer_grads_rows = 0) - and the deferred steps replay as plans."""
    from midi_vae_amd import ops
    B = 48
    spec, params, batch, raw = _problem(cell, B, seed=61, H=256, Z=64, T=64)
    res = {}
    for rows in (32768, 32767, 0):
        eng = Engine(spec, max_batch=B, dtype="bf16")
        eng.defer_grads_rows = rows
        eng.defer_early, eng.defer_early_rows = rows == 32768, 0
        eng.set_params(params)
        _stage(eng, raw, B)
        launched, real = [], ops.gemm_multi
        ops.gemm_multi = lambda problems, stream=None: (launched.append(len(problems)), real(problems, stream=stream))[1]
        try:
            eng.forward_backward(B)
        finally:
            ops.gemm_multi = real
        assert ((len(launched) == (2 if eng.defer_early else 1) and sum(launched) >= 8 and min(launched) >= 3) if rows
                else not launched), launched
        m0, g0 = eng.metrics(B), eng.get_grads()
        losses = []
        for _ in range(8):
            eng.train_step(B)
            losses.append(eng.metrics(B)["loss"])
        eng.check_pipeline()
        res[rows] = (m0, g0, losses, dict(eng.plan_stats))
    m0, g0, l0, _ = res[0]
    for rows in (32768, 32767):
        m1, g1, l1, st1 = res[rows]
        assert st1["replayed"] >= 2, st1
        for k in m0:
            assert m1[k] == pytest.approx(m0[k], rel=1e-6, abs=1e-7), k
        for k in g0:
            assert np.linalg.norm(g1[k] - g0[k]) <= 1e-4 * (np.linalg.norm(g0[k]) + 1e-8) + 1e-7, k
        for a, b in zip(l1, l0):
            assert abs(a - b) <= 1e-4 * (1 + abs(b)), (l1, l0)


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_ragged_minibatches_of_one_padded_size_share_a_plan(cell):
    """Engine._kind_B (round 5): on the default graph a train step's launch list depends on the PADDED batch only; the real number
    of windows and 1 / (global minibatch size) are call parameters (ops.ParamInt / ParamFloat) that a replayed plan patches into
    the latent chain and the heads.  Songs of 100, 97, 104, 99 ... windows (all padded to 112) - what `python vae_training.py`
    feeds at the reference's default settings - then replay one plan: per-step losses and the final parameters equal those of an
    engine that enqueues every step from Python, and most steps are replays."""
    sizes = [100, 97, 104, 99, 101, 98, 103, 112, 105, 97, 110, 100]
    spec, params, batch, raw = _problem(cell, 112, seed=71, H=256, Z=64, T=64)
    res = {}
    for plans in (True, False):
        eng = Engine(spec, max_batch=112, dtype="bf16")
        eng.use_plans = plans
        eng.set_params(params)
        losses = []
        for B in sizes:
            sub = {k: (v[:B] if hasattr(v, "shape") and v.shape[:1] == (112,) else v) for k, v in raw.items()}
            _stage(eng, sub, B)
            eng.train_step(B)
            losses.append(eng.metrics(B)["loss"])
        eng.check_pipeline()
        res[plans] = (losses, eng.get_params(), dict(eng.plan_stats))
    assert res[True][2]["replayed"] >= len(sizes) - 6, res[True][2]
    assert len(set(round(l, 4) for l in res[False][0])) > 6          # (the steps really differ)
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 2e-4 * (1 + abs(b)), (res[True][0], res[False][0])
    for k, v in res[False][1].items():
        assert np.linalg.norm(res[True][1][k] - v) <= 2e-3 * (np.linalg.norm(v) + 1e-6) + 1e-6, k


def test_a_refused_latent_chain_keys_plans_by_the_exact_window_count():
    """when the library refuses the fused latent chain (MVAE_E_UNSUPPORTED: more than 160 KB of LDS) the step runs the
    separate latent launches, which take the real window count and the loss normaliser as PLAIN arguments and zero the padding
    rows with a torch operation only on a ragged batch.  Such steps must not share a plan per padded size: three full minibatches
    would arm one that a ragged minibatch then replays with a stale count.  Forced here by refusing the chain from Python; the
    losses of a mixed sequence of window counts equal those of an engine that never replays."""
    sizes = [112, 112, 112, 112, 100, 97, 112, 100, 97, 100, 97]
    spec, params, batch, raw = _problem("GRU", 112, seed=73, H=256, Z=64, T=32)
    res = {}
    for plans in (True, False):
        eng = Engine(spec, max_batch=112, dtype="bf16")
        eng.use_plans = plans
        eng.set_params(params)
        eng._latent_chain_forward = lambda *a, **k: False
        eng._latent_chain_backward = lambda *a, **k: None
        losses = []
        for B in sizes:
            sub = {k: (v[:B] if hasattr(v, "shape") and v.shape[:1] == (112,) else v) for k, v in raw.items()}
            _stage(eng, sub, B)
            eng.train_step(B)
            losses.append(eng.metrics(B)["loss"])
        eng.check_pipeline()
        assert eng._chain_refused
        assert eng._kind_B(100) == (100, float(eng.norm_B))
        res[plans] = losses
    for a, b in zip(res[True], res[False]):
        assert abs(a - b) <= 2e-4 * (1 + abs(b)), (res[True], res[False])
