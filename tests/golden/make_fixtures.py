#!/usr/bin/env python3
"""Generate golden vectors from the importable (pure-NumPy) part of the reference.

Runs ONLY in the build container (needs /root/reference).  The reference's Keras /
recurrentshop / pretty_midi imports are replaced by throw-away stub modules so that the
NumPy helpers of vae_definition.py / midi_functions.py / data_class.py can be called on
seeded inputs.  Only the INPUTS and OUTPUTS (data) are written to tests/golden/*.npz;
no reference source travels.

  python tests/golden/make_fixtures.py            # rewrites tests/golden/*.npz

Pinned functions (reference file:line):
  vae_definition.py:770-808   prepare_encoder_input_list
  vae_definition.py:816-865   prepare_decoder_input
  vae_definition.py:880-1045  prepare_autoencoder_input_and_output_list
  vae_definition.py:1048-1067 sample_vector('argmax')
  vae_definition.py:1071-1095 sample_notes_prediction
  vae_definition.py:1097-1107 sample_instrument_prediction
  vae_definition.py:1131-1225 process_decoder_outputs('argmax')
  midi_functions.py:14-54     programs_to_instrument_matrix
  data_class.py:241-252       monophonic_to_khot_pianoroll
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub_modules():
    class _Dummy(object):
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Dummy()

    def to_categorical(y, num_classes=None):
        y = np.array(y, dtype="int").ravel()
        if not num_classes:
            num_classes = np.max(y) + 1
        out = np.zeros((y.shape[0], num_classes))
        out[np.arange(y.shape[0]), y] = 1
        return out

    names = {
        "keras": ["objectives", "backend"],
        "keras.layers": ["Bidirectional", "Dense", "Embedding", "Input", "Lambda", "LSTM", "RepeatVector",
                         "TimeDistributed", "Add", "GRU", "SimpleRNN", "Layer"],
        "keras.models": ["Model"],
        "keras.layers.merge": ["Concatenate"],
        "keras.utils": [],
        "recurrentshop": ["RecurrentModel", "Activation"],
        "recurrentshop.cells": ["LSTMCell", "GRUCell", "SimpleRNNCell"],
        "matplotlib2tikz": ["save"],
        "pretty_midi": [],
        "mido": [],
    }
    for mod, attrs in names.items():
        m = types.ModuleType(mod)
        for a in attrs:
            setattr(m, a, _Dummy)
        m.__all__ = list(attrs)
        sys.modules[mod] = m
    sys.modules["keras.utils"].to_categorical = to_categorical
    sys.modules["keras"].optimizers = types.SimpleNamespace(Adam=_Dummy, RMSprop=_Dummy)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; fixtures can only be regenerated in the build container")
    _stub_modules()
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="mvae_fixture_")
    os.chdir(tmp)  # settings.py makes pickles/<t>/ in the CWD at import (settings.py:58-61)
    sys.path.insert(0, REF)
    import settings  # noqa: F401
    import vae_definition as vd
    import midi_functions as mf
    import data_class as dc
    os.chdir(cwd)

    rng = np.random.default_rng(20240928)
    T = settings.output_length          # 64
    D = settings.output_dim             # 61
    V = settings.max_voices             # 4
    ID = settings.meta_instrument_dim   # 16
    Z = settings.latent_dim             # 256
    n = 5

    def roll(nw):
        idx = np.where(rng.random((nw, T)) < 0.35, D - 1, rng.integers(0, D - 1, (nw, T)))
        X = np.zeros((nw, T, D))
        X[np.arange(nw)[:, None], np.arange(T)[None, :], idx] = 1
        return X, idx

    X, xidx = roll(n)
    Y = X.copy()
    vel = np.where((xidx == D - 1) | (rng.random((n, T)) < 0.5), 0.0, 0.5 + 0.5 * rng.random((n, T)))
    held = (rng.random((n, T)) < 0.3).astype(np.float64)
    progs = [0, 33, 48, 127]
    I = mf.programs_to_instrument_matrix(progs, settings.instrument_attach_method, V)
    S = rng.standard_normal((n, settings.signature_vector_length))
    H = rng.standard_normal((n, Z)) * 0.1
    R = rng.standard_normal((n, Z))
    C = 1

    out = {"X": X, "Y": Y, "V": vel, "D": held, "I": I, "S": S, "H": H, "R": R, "C": np.int64(C),
           "programs": np.array(progs)}

    enc = vd.prepare_encoder_input_list(X, I, vel, held)
    for i, a in enumerate(enc):
        out["enc_in_%d" % i] = np.asarray(a)
    out["enc_in_n"] = np.int64(len(enc))

    dec = vd.prepare_decoder_input(R, C, S, H)
    for i, a in enumerate(dec):
        out["dec_in_%d" % i] = np.asarray(a)
    out["dec_in_n"] = np.int64(len(dec))
    dec2 = vd.prepare_decoder_input(R, C, S, None)
    for i, a in enumerate(dec2):
        out["dec_in_autoH_%d" % i] = np.asarray(a)

    xi, yo, sw = vd.prepare_autoencoder_input_and_output_list(X, Y, C, I, vel, held, S, H, return_sample_weight=True)
    for i, a in enumerate(xi):
        out["ae_x_%d" % i] = np.asarray(a)
    for i, a in enumerate(yo):
        out["ae_y_%d" % i] = np.asarray(a)
    for i, a in enumerate(sw):
        out["ae_w_%d" % i] = np.asarray(a)
    out["ae_x_n"] = np.int64(len(xi))
    out["ae_y_n"] = np.int64(len(yo))
    out["ae_w_n"] = np.int64(len(sw))
    xi2, yo2 = vd.prepare_autoencoder_input_and_output_list(X, Y, C, I, vel, held, S, H, return_sample_weight=False)
    assert len(xi2) == len(xi) and len(yo2) == len(yo)

    # ---- argmax decode ----------------------------------------------------------------------
    probs = rng.random((n, T, D))
    probs /= probs.sum(-1, keepdims=True)
    probs[0, 3, :] = 0.0                    # all-zero row -> index 0 (vae_definition.py:1049,1065-1066)
    probs[1, 5, :] = 0.0
    probs[1, 5, D - 1] = 1.0                # silent wins -> all-zero output row (:1090-1091)
    probs[2, 7, 10] = probs[2, 7, 20] = 0.4  # tie -> first max
    probs[2, 7, 30:] = 0.0
    probs[2, 7, :10] = 0.0
    probs[2, 7, 11:20] = 0.0
    probs[2, 7, 21:30] = 0.0
    pinstr = rng.random((n, V, ID))
    pinstr /= pinstr.sum(-1, keepdims=True)
    pvel = rng.random((n, T, 1))
    out["probs_notes"] = probs
    out["probs_instr"] = pinstr
    out["pred_vel"] = pvel
    out["notes_argmax_3d"] = vd.sample_notes_prediction(probs, "argmax")
    out["notes_argmax_2d"] = vd.sample_notes_prediction(probs[0], "argmax")
    out["instr_argmax"] = vd.sample_instrument_prediction(pinstr, "argmax")
    out["held_argmax"] = vd.sample_held_notes_prediction(rng.random((n, T, 2)) * 0 + np.stack(
        [probs[..., 0], probs[..., 1]], -1), "argmax")
    out["held_probs"] = np.stack([probs[..., 0], probs[..., 1]], -1)
    Yd, Id, Vd, Dd, Nd = vd.process_decoder_outputs([probs, pinstr, pvel], "argmax")
    out["proc_Y"], out["proc_I"], out["proc_V"], out["proc_D"], out["proc_N"] = Yd, Id, Vd, Dd, Nd
    Yb, Ib, Vb, Db, Nb = vd.process_decoder_outputs(probs, "argmax")  # bare array (no meta heads) branch
    out["procb_Y"], out["procb_I"], out["procb_V"], out["procb_D"], out["procb_N"] = Yb, Ib, Vb, Db, Nb

    # ---- instrument matrices / k-hot rolls ----------------------------------------------------
    for m in ("1hot-category", "khot-category", "1hot-instrument", "khot-instrument"):
        out["instr_" + m] = mf.programs_to_instrument_matrix(progs, m, V)
    mono = Y[0][:, :D - 1]
    out["khot_in"] = mono
    out["khot_out"] = dc.monophonic_to_khot_pianoroll(mono, V)
    out["khot_out_raw"] = dc.monophonic_to_khot_pianoroll(mono, V, set_all_nonzero_to_1=False)

    # ---- the settings surface (names + scalar values) -----------------------------------------
    sett = {}
    for k in sorted(vars(settings)):
        v = getattr(settings, k)
        if k.startswith("_") or isinstance(v, types.ModuleType):
            continue
        if k in ("t", "pickle_store_folder", "instrument_names"):
            continue  # run-specific values / the 128-entry GM name table (plots only, out of scope)
        if isinstance(v, (bool, int, float, str)) or v is None:
            sett[k] = v
        elif isinstance(v, (list, tuple)) and all(isinstance(e, (str, int, float)) for e in v):
            sett[k] = list(v)
    import json
    with open(os.path.join(HERE, "settings_surface.json"), "w") as f:
        json.dump(sett, f, indent=1, sort_keys=True)

    np.savez_compressed(os.path.join(HERE, "packers_decode.npz"), **out)
    print("wrote", os.path.join(HERE, "packers_decode.npz"), "keys:", len(out))
    print("wrote settings_surface.json keys:", len(sett))


if __name__ == "__main__":
    main()
