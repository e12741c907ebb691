"""Host packers / argmax decode vs golden vectors captured from the reference's own NumPy helpers
(tests/golden/make_fixtures.py; reference vae_definition.py:770-1235, midi_functions.py:14-54,
data_class.py:241-252).  Bit-exact: these are integer / one-hot / copy operations."""
import json
import os

import numpy as np
import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd import packers as pk
from midi_vae_amd.config import build_settings, create_kwargs

HERE = os.path.dirname(os.path.abspath(__file__))


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b)


def test_settings_surface_matches_reference():
    ref = json.load(open(os.path.join(HERE, "golden", "settings_surface.json")))
    s = build_settings()
    for k, v in ref.items():
        assert k in s, k
        assert s[k] == v, (k, s[k], v)


def test_root_settings_module_is_flat_namespace_without_side_effects(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    import importlib
    import settings
    importlib.reload(settings)
    assert settings.output_length == 64 and settings.input_dim == 61 and settings.cell_type == "GRU"
    assert not os.path.exists(tmp_path / "pickles")      # reference settings.py:58-61 would create it


def test_settings_override_and_derivation():
    s = build_settings(cell_type="LSTM", latent_dim=64, input_length=128, output_length=128, max_voices=4,
                       classes=("C", "J", "P", "B"))
    assert s["output_length"] == 512 and s["input_length"] == 512 and s["num_classes"] == 4
    assert s["num_composers"] == 4 and s["meta_velocity_length"] == 512
    with pytest.raises(KeyError):
        build_settings(not_a_knob=1)
    kw = create_kwargs(s)
    assert len(kw) == 61 and kw["latent_rep_size"] == 64 and kw["cell_type"] == "LSTM"


def test_prepare_encoder_input_list(golden, default_settings):
    out = pk.prepare_encoder_input_list(default_settings, golden["X"], golden["I"], golden["V"], golden["D"])
    assert len(out) == int(golden["enc_in_n"])
    for i, a in enumerate(out):
        _eq(a, golden["enc_in_%d" % i])


def test_prepare_decoder_input(golden, default_settings):
    out = pk.prepare_decoder_input(default_settings, golden["R"], int(golden["C"]), golden["S"], golden["H"])
    assert len(out) == int(golden["dec_in_n"])
    for i, a in enumerate(out):
        _eq(a, golden["dec_in_%d" % i])
    out = pk.prepare_decoder_input(default_settings, golden["R"], int(golden["C"]), golden["S"], None)
    for i, a in enumerate(out):
        _eq(a, golden["dec_in_autoH_%d" % i])


def test_prepare_autoencoder_lists(golden, default_settings):
    g = golden
    x, y, w = pk.prepare_autoencoder_input_and_output_list(default_settings, g["X"], g["Y"], int(g["C"]), g["I"],
                                                           g["V"], g["D"], g["S"], g["H"], return_sample_weight=True)
    assert (len(x), len(y), len(w)) == (int(g["ae_x_n"]), int(g["ae_y_n"]), int(g["ae_w_n"]))
    for i, a in enumerate(x):
        _eq(a, g["ae_x_%d" % i])
    for i, a in enumerate(y):
        _eq(a, g["ae_y_%d" % i])
    for i, a in enumerate(w):
        _eq(a, g["ae_w_%d" % i])
    x2, y2 = pk.prepare_autoencoder_input_and_output_list(default_settings, g["X"], g["Y"], int(g["C"]), g["I"],
                                                          g["V"], g["D"], g["S"], g["H"])
    assert len(x2) == len(x) and len(y2) == len(y)
    # caller's arrays are never mutated (reference copies V, vae_definition.py:783,894)
    _eq(g["V"], np.load(os.path.join(HERE, "golden", "packers_decode.npz"))["V"])


def test_argmax_decode(golden, default_settings):
    g, s = golden, default_settings
    _eq(pk.sample_notes_prediction(s, g["probs_notes"], "argmax"), g["notes_argmax_3d"])
    _eq(pk.sample_notes_prediction(s, g["probs_notes"][0], "argmax"), g["notes_argmax_2d"])
    _eq(pk.sample_instrument_prediction(s, g["probs_instr"], "argmax"), g["instr_argmax"])
    _eq(pk.sample_held_notes_prediction(s, g["held_probs"], "argmax"), g["held_argmax"])
    # edge cases planted by the generator
    idx = pk.note_indices(s, g["probs_notes"], "argmax").reshape(g["probs_notes"].shape[:2])
    assert idx[0, 3] == 0          # all-zero row -> index 0
    assert idx[1, 5] == 60         # silent class wins -> all-zero output row
    assert idx[2, 7] == 10         # tie -> first maximum
    assert g["notes_argmax_3d"][1 * 64 + 5].sum() == 0


def test_process_decoder_outputs(golden, default_settings):
    g, s = golden, default_settings
    out = pk.process_decoder_outputs(s, [g["probs_notes"], g["probs_instr"], g["pred_vel"]], "argmax")
    for a, k in zip(out, ("proc_Y", "proc_I", "proc_V", "proc_D", "proc_N")):
        _eq(a, g[k])
    out = pk.process_autoencoder_outputs(s, g["probs_notes"], "argmax")
    for a, k in zip(out, ("procb_Y", "procb_I", "procb_V", "procb_D", "procb_N")):
        _eq(a, g[k])


def test_instrument_matrices_and_khot(golden):
    g = golden
    for m in ("1hot-category", "khot-category", "1hot-instrument", "khot-instrument"):
        _eq(pk.programs_to_instrument_matrix(g["programs"], m, 4), g["instr_" + m])
    with pytest.raises(ValueError):
        pk.programs_to_instrument_matrix([0], "nope", 4)
    _eq(pk.monophonic_to_khot_pianoroll(g["khot_in"], 4), g["khot_out"])
    _eq(pk.monophonic_to_khot_pianoroll(g["khot_in"], 4, set_all_nonzero_to_1=False), g["khot_out_raw"])


def test_empty_and_ragged_inputs(default_settings):
    s = default_settings
    X = np.zeros((0, 64, 61))
    out = pk.prepare_encoder_input_list(s, X, np.zeros((4, 16)), np.zeros((0, 64)), np.zeros((0, 64)))
    assert [a.shape for a in out] == [(0, 64, 61), (0, 4, 16), (0, 64, 1)]
    assert pk.sample_notes_prediction(s, np.zeros((0, 64, 61)), "argmax").shape == (0, 60)
    one = pk.prepare_decoder_input(s, np.ones((1, 256)), 0, None, None)
    assert np.all(one[2] == 0)     # a single window has zero history


def test_onehot_to_index_roundtrip(golden):
    idx = pk.onehot_to_index(golden["X"])
    assert idx.dtype == np.uint8 and idx.shape == golden["X"].shape[:2]
    back = np.zeros(golden["X"].shape)
    np.put_along_axis(back, idx[..., None].astype(np.int64), 1, -1)
    _eq(back, golden["X"])
