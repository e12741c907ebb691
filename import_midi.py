"""Drop-in for the reference's ``import_midi`` module - the part of it that is on this repo's side of the scope line.

reference import_midi.py:352-574 ``import_midi_from_folder(folder)`` returns sixteen per-song lists
    V_train, V_test, D_train, D_test, T_train, T_test, I_train, I_test, Y_train, Y_test, X_train, X_test, c_train, c_test,
    train_paths, test_paths
(V velocity rolls (n_win, T), D held-notes rolls (n_win, T), T tempi, I instrument matrices (max_voices, 16), Y / X one-hot window
tensors (n_win, T, 61), c class indices, paths).  With ``load_from_pickle_instead_of_midi`` it reads them from sixteen
``<name>.pickle`` files under ``pickle_load_path`` (:355-373); after importing MIDI files it writes the same files to
``pickle_store_folder`` (:548-571).  That cache format is reproduced here, both ways, so that a dataset imported once with the
reference feeds ``vae_training.py`` / ``style_classifier_training.py`` unchanged.  Parsing MIDI files themselves (pretty_midi,
:13-250) is outside the hot path and impossible in this image (pretty_midi is absent): asked for, it raises.  The ARRAY-level end of
``load_rolls`` - silent column, right padding, the split into windows (:256-265, :303-345) - needs no MIDI library and is here as
``windows_from_unrolled_rolls``: it is also what turns a decoded song (``process_decoder_outputs``: rows without the silent column,
all-zero where the model chose silence, ``vae_definition.py:1084-1093``) back into windows the encoder takes.
"""
import os
import pickle

import numpy as np

import settings

NAMES = ("V_train", "V_test", "D_train", "D_test", "T_train", "T_test", "I_train", "I_test", "Y_train", "Y_test", "X_train",
         "X_test", "c_train", "c_test", "train_paths", "test_paths")


def load_pickle_cache(path):
    """the sixteen lists, in the reference's return order (reference import_midi.py:355-373)"""
    out = []
    for name in NAMES:
        with open(os.path.join(path, name + ".pickle"), "rb") as f:
            out.append(pickle.load(f))
    return tuple(out)


def save_pickle_cache(path, lists):
    """the reference's cache files (reference import_midi.py:548-571); ``lists`` in the order of NAMES"""
    if len(lists) != len(NAMES):
        raise ValueError("expected the %d lists %s" % (len(NAMES), NAMES))
    os.makedirs(path, exist_ok=True)
    for name, obj in zip(NAMES, lists):
        with open(os.path.join(path, name + ".pickle"), "wb") as f:
            pickle.dump(obj, f)


def _split_padded(a, length, silent_fill):
    """right-pad the first axis to a multiple of ``length`` and split (reference import_midi.py:307-345)"""
    a = np.asarray(a, dtype=np.float64)
    pad = (-a.shape[0]) % length
    a = np.pad(a, ((0, pad),) + ((0, 0),) * (a.ndim - 1), "constant")
    if silent_fill and pad:
        a[a.shape[0] - pad:, -1] = 1
    return a.reshape((a.shape[0] // length, length) + a.shape[1:])


def windows_from_unrolled_rolls(Y, V, D, s=None, as_written=False):
    """(X, Y, V, D) window arrays from one song's UNROLLED rolls - the array-level end of the reference's ``load_rolls``.

    ``Y``: (song_length, high_crop - low_crop) rows with at most one 1 (voices interleaved step by step, already cropped);
    ``V``, ``D``: (song_length,) velocity / held-note rolls.  As reference import_midi.py:256-265 a silent-note column is appended
    and set where a row is empty (``include_silent_note``), :296-300 ``X`` is ``Y`` (every ``max_voices``-th row under
    ``song_completion``), :303-345 everything is padded on the right to whole windows - padding rows are silent notes, velocity 0,
    not held - and split into (n, input_length, .) / (n, output_length, .).

    ``as_written``: the reference writes the padding rows' silent bit as ``X[-padding_length:, -1] = 1``; for a song that needs NO
    padding that slice is ``X[-0:]`` = every row, and the whole song gets the silent bit on top of its notes.  Off by default
    (those rows are not one-hot and the engine's staging refuses them); on, the arrays equal the reference's for such songs too."""
    s = vars(settings) if s is None else s
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim != 2 or Y.shape[1] != s["high_crop"] - s["low_crop"]:
        raise ValueError("Y must be (song_length, %d), got %r" % (s["high_crop"] - s["low_crop"], Y.shape))
    if np.any(Y.sum(1) > 1):
        raise ValueError("more than one note in a row: the rolls must be unrolled voice by voice (reference import_midi.py:251-252)")
    silent = bool(s["include_silent_note"])
    if silent:
        Y = np.concatenate([Y, (Y.sum(1) == 0)[:, None].astype(np.float64)], axis=1)
    X = Y[::s["max_voices"]] if s["song_completion"] else Y

    def split(a, length):
        out = _split_padded(a, length, silent and a.ndim == 2)
        if as_written and silent and a.ndim == 2 and a.shape[0] % length == 0:
            out[..., -1] = 1
        return out

    Lin, Lout = s["input_length"], s["output_length"]
    Xw = split(X, Lin) if Lin > 0 else X
    if Lout > 0:
        return Xw, split(Y, Lout), split(V, Lout), split(D, Lout)
    return Xw, Y, np.asarray(V, dtype=np.float64), np.asarray(D, dtype=np.float64)


def import_midi_from_folder(folder):
    s = vars(settings)
    if s.get("load_from_pickle_instead_of_midi"):
        return load_pickle_cache(s["pickle_load_path"])
    raise NotImplementedError(
        "importing MIDI files (%r) needs pretty_midi, which this image does not have, and is outside this repo's scope (SURVEY "
        "section 2: MIDI import / export); import the dataset once with the reference (save_imported_midi_as_pickle = True) and set "
        "load_from_pickle_instead_of_midi = True / pickle_load_path - or pass --pickle-dir to vae_training.py" % (folder,))
