"""Drop-in for the reference's ``import_midi`` module - the part of it that is on this repo's side of the scope line.

reference import_midi.py:352-574 ``import_midi_from_folder(folder)`` returns sixteen per-song lists
    V_train, V_test, D_train, D_test, T_train, T_test, I_train, I_test, Y_train, Y_test, X_train, X_test, c_train, c_test,
    train_paths, test_paths
(V velocity rolls (n_win, T), D held-notes rolls (n_win, T), T tempi, I instrument matrices (max_voices, 16), Y / X one-hot window
tensors (n_win, T, 61), c class indices, paths).  With ``load_from_pickle_instead_of_midi`` it reads them from sixteen
``<name>.pickle`` files under ``pickle_load_path`` (:355-373); after importing MIDI files it writes the same files to
``pickle_store_folder`` (:548-571).  That cache format is reproduced here, both ways, so that a dataset imported once with the
reference feeds ``vae_training.py`` / ``style_classifier_training.py`` unchanged.  Parsing MIDI files themselves (pretty_midi,
:13-350) is outside the hot path and impossible in this image (pretty_midi is absent): asked for, it raises.
"""
import os
import pickle

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


def import_midi_from_folder(folder):
    s = vars(settings)
    if s.get("load_from_pickle_instead_of_midi"):
        return load_pickle_cache(s["pickle_load_path"])
    raise NotImplementedError(
        "importing MIDI files (%r) needs pretty_midi, which this image does not have, and is outside this repo's scope (SURVEY "
        "section 2: MIDI import / export); import the dataset once with the reference (save_imported_midi_as_pickle = True) and set "
        "load_from_pickle_instead_of_midi = True / pickle_load_path - or pass --pickle-dir to vae_training.py" % (folder,))
