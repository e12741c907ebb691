"""The drop-in Python surface (VAE.create / model views / root modules) - everything that needs no GPU."""
import os

import numpy as np
import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd.config import build_settings, create_kwargs
from midi_vae_amd.layout import ModelSpec, ParamLayout, dec_init_blocks, init_params, spec_from_create_kwargs
from midi_vae_amd.model import VAE
from oracle.vae_oracle import make_cfg, param_shapes


def _kw(**over):
    return create_kwargs(build_settings(**over))


def test_create_accepts_the_reference_call_and_exposes_three_models():
    m = VAE().create(**_kw())
    assert m.encoder is not None and m.decoder is not None and m.autoencoder is not None
    assert m.lstm_size == 256 and m.cell_type == "GRU"          # attributes mirrored like vae_definition.py:109-172
    assert m.autoencoder.metrics_names == ["loss", "decoder_loss", "decoder_loss", "decoder_loss",
                                           "composer_decoder_loss", "decoder_acc", "decoder_acc", "decoder_acc",
                                           "composer_decoder_acc"]
    assert m.autoencoder.count_params() == 2966094             # matches SURVEY section 8 "reference default 2.97 M"
    assert "Total params" in m.encoder.summary()
    m.autoencoder.reset_states()


def test_history_keys_are_the_ones_the_training_script_reads():
    m = VAE().create(**_kw())
    keys = [k for k, _ in m.autoencoder._history_keys()]
    for k in ("loss", "decoder_loss_1", "decoder_acc_1", "decoder_loss_2", "decoder_acc_2", "decoder_loss_3",
              "decoder_acc_3", "composer_decoder_loss", "composer_decoder_acc"):     # vae_training.py:817-864
        assert k in keys
    m2 = VAE().create(**_kw(meta_instrument=False, meta_velocity=False, include_composer_decoder=False))
    assert [k for k, _ in m2.autoencoder._history_keys()] == ["loss", "acc"]
    assert m2.autoencoder.metrics_names == ["loss", "acc"]


@pytest.mark.parametrize("switch", ["use_embedding"])
def test_unimplemented_switches_fail_loudly(switch):
    kw = _kw()
    kw[switch] = True
    with pytest.raises((NotImplementedError, AssertionError)):
        VAE().create(**kw)


def test_asserts_of_the_reference_are_kept():
    kw = _kw()
    kw["beta"] = 0
    with pytest.raises(AssertionError):
        VAE().create(**kw)
    kw = _kw()
    kw["lstm_size"] = 100
    with pytest.raises(NotImplementedError):
        VAE().create(**kw)


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "SimpleRNN"])
def test_layout_agrees_with_oracle_naming_and_roundtrips(cell):
    spec = spec_from_create_kwargs(_kw(cell_type=cell, latent_dim=64))
    L = ParamLayout.build(spec)
    o = param_shapes(make_cfg(**spec.oracle_cfg()))
    assert set(o) == set(L.oracle_names())
    for k, shp in o.items():
        assert tuple(shp) == L.entries[k].shape, k
    p = init_params(spec, 3)
    back = L.unpack(L.pack(p))
    assert all(np.array_equal(p[k], back[k]) for k in p)
    # decoder-side tensors form one contiguous bucket at the end (first gradient all-reduce bucket)
    assert all((L.entries[n].offset >= L.dec_begin) == n.startswith("dec.") for n in L.entries)
    # initial-state Denses are column blocks of one matrix
    nb = len(dec_init_blocks(spec))
    assert L.entries["dec.init.W"].shape == (spec.zin, nb * spec.H)
    assert L.entries["dec.notes.init.0.0.W"].row_stride == nb * spec.H


def test_initialisers():
    spec = ModelSpec(cell="LSTM", H=64, Z=32)
    p = init_params(spec, 0)
    U = p["enc.notes.0.U"].astype(np.float64)
    assert np.allclose(U @ U.T, np.eye(64), atol=1e-5)                 # orthogonal recurrent kernels
    lim = np.sqrt(6.0 / (61 + 4 * 64))
    assert np.abs(p["enc.notes.0.W"]).max() <= lim + 1e-7              # glorot_uniform over the concatenated width
    assert np.all(p["enc.notes.0.b"][64:128] == 1) and p["enc.notes.0.b"][:64].sum() == 0    # Keras unit_forget_bias
    assert np.all(p["dec.notes.0.b"] == 0)                             # recurrentshop cells: plain zero bias


def test_save_and_load_weights_roundtrip(tmp_path):
    m = VAE().create(**_kw(latent_dim=64))
    for view, name in ((m.autoencoder, "autoencoderEpoch10.pickle"), (m.encoder, "encoderEpoch10.pickle"),
                       (m.decoder, "decoderEpoch10.pickle")):          # file names of vae_training.py:966-978
        view.save_weights(str(tmp_path / name))
    m2 = VAE().create(seed=123, **_kw(latent_dim=64))
    before = m2.autoencoder.get_weights()
    m2.encoder.load_weights(str(tmp_path / "encoderEpoch10.pickle"))
    m2.decoder.load_weights(str(tmp_path / "decoderEpoch10.pickle"), by_name=False)
    after, ref = m2.autoencoder.get_weights(), m.autoencoder.get_weights()
    assert any(not np.array_equal(a, b) for a, b in zip(before, after))
    assert all(np.array_equal(a, b) for a, b in zip(after, ref))
    m3 = VAE().create(**_kw(latent_dim=32))
    with pytest.raises(ValueError):
        m3.autoencoder.load_weights(str(tmp_path / "autoencoderEpoch10.pickle"))


def test_root_modules_are_drop_in(golden):
    import settings
    import vae_definition as vd
    out = vd.prepare_encoder_input_list(golden["X"], golden["I"], golden["V"], golden["D"])
    assert [a.shape for a in out] == [(5, 64, 61), (5, 4, 16), (5, 64, 1)]
    x, y, w = vd.prepare_autoencoder_input_and_output_list(golden["X"], golden["Y"], 1, golden["I"], golden["V"],
                                                           golden["D"], golden["S"], golden["H"], return_sample_weight=True)
    assert len(x) == 7 and len(y) == 4 and len(w) == 4
    assert vd.sample_notes_prediction(golden["probs_notes"], "argmax").shape == (5 * 64, 60)
    assert hasattr(vd, "VAE") and settings.output_length == 64


def test_non_onehot_input_is_rejected_not_silently_accepted():
    from midi_vae_amd.staging import host_onehot_to_index as _to_index
    with pytest.raises(NotImplementedError):
        _to_index(np.full((2, 3, 4), 0.25), "notes input")
    assert _to_index(np.eye(4)[None], "x").tolist() == [[0, 1, 2, 3]]


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from midi_vae_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(ModelSpec(H=64, Z=32), max_batch=16)


def test_first_use_verification_retries_once_then_falls_back():
    """Engine._verify_pipeline (host logic only): a timed-out first use of a kind of call is redone once as it is; only a second
    time-out switches the engine to chunked launches; every kind of call is verified separately - by its first call that
    actually ran a stack time-pipelined."""
    import types
    import warnings
    import torch
    from midi_vae_amd.engine import Engine

    def fake(status_after_redo):
        calls = []
        eng = types.SimpleNamespace(pipeline=True, _pipe_used=True, _pipe_verified=set(), _dxp0_clean=True,
                                    store={"pipe_status": torch.tensor([3], dtype=torch.int32)})

        def redo():
            calls.append(1)
            eng.store["pipe_status"].fill_(status_after_redo[min(len(calls), len(status_after_redo)) - 1])
        return eng, redo, calls

    eng, redo, calls = fake([0])                     # transient: the retry succeeds
    Engine._verify_pipeline(eng, redo, key="train")
    assert calls == [1] and eng.pipeline and "train" in eng._pipe_verified and int(eng.store["pipe_status"]) == 0
    Engine._verify_pipeline(eng, redo, key="train")  # verified once per kind
    assert calls == [1]
    eng.store["pipe_status"].fill_(2)
    Engine._verify_pipeline(eng, redo, key="decode")  # another kind of call is checked on ITS first use
    assert calls == [1, 1] and eng.pipeline

    eng, redo, calls = fake([0])                     # a call whose batch ran nothing pipelined verifies nothing
    eng._pipe_used = False
    Engine._verify_pipeline(eng, redo, key="train")
    assert calls == [] and "train" not in eng._pipe_verified

    eng, redo, calls = fake([4, 0])                  # persistent: falls back, redoes once more
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Engine._verify_pipeline(eng, redo, key="train")
    assert calls == [1, 1] and not eng.pipeline and len(w) == 1
    assert int(eng.store["pipe_status"]) == 0 and eng._dxp0_clean is False


def test_epsilon_stream_is_one_stream_notices_a_moved_generator_and_ends_its_thread():
    """model.EpsilonStream (the sampling layer's N(0, epsilon_std) draws, reference vae_definition.py:498-502) draws ahead of its
    consumer on a worker thread: rows come out in the order successive draws would give them; a generator the caller reseeds or
    restores in place between two draws is noticed (the buffered rows are dropped); a replaced stream's thread ends."""
    import threading
    import time
    from midi_vae_amd.model import EpsilonStream
    rng = np.random.default_rng(3)
    st = EpsilonStream(rng, 8, 0.5)
    full = (np.random.default_rng(3).standard_normal((1024, 8)) * 0.5).astype(np.float32)
    assert np.array_equal(st.take(5), full[:5]) and np.array_equal(st.take(7), full[5:12])
    for _ in range(200):                       # (the block drawn ahead, if any, finishes)
        if st._next is None or st._next.done():
            break
        time.sleep(0.01)
    rng.bit_generator.state = np.random.default_rng(9).bit_generator.state
    fresh = (np.random.default_rng(9).standard_normal((1024, 8)) * 0.5).astype(np.float32)
    assert np.array_equal(st.take(4), fresh[:4])
    st.reset()                                 # explicit: the rows drawn ahead are forgotten, the generator is where it is
    before = threading.active_count()
    st.close()
    assert threading.active_count() == before - 1
