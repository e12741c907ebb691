"""The reference's dataset cache (import_midi.py:355-373 read, :548-571 write): sixteen pickles of per-song lists, round trip and
through the ``load_from_pickle_instead_of_midi`` switch of ``import_midi_from_folder``; MIDI parsing itself is refused loudly."""
import numpy as np
import pytest

import import_midi
import settings
import vae_training


def _lists(rng, n_train=3, n_test=2, T=16):
    def songs(k):
        X = [np.eye(61)[rng.integers(0, 61, (int(rng.integers(2, 6)), T))] for _ in range(k)]
        V = [rng.random(x.shape[:2]) for x in X]
        D = [(rng.random(x.shape[:2]) < 0.3).astype(float) for x in X]
        I = [np.eye(16)[rng.integers(0, 16, (4,))] for _ in X]
        return X, V, D, I, [120.0] * k, [int(i % 2) for i in range(k)], ["song%d.mid" % i for i in range(k)]
    Xa, Va, Da, Ia, Ta, ca, pa = songs(n_train)
    Xb, Vb, Db, Ib, Tb, cb, pb = songs(n_test)
    return (Va, Vb, Da, Db, Ta, Tb, Ia, Ib, Xa, Xb, Xa, Xb, ca, cb, pa, pb)


def test_pickle_cache_round_trip_and_switch(tmp_path, monkeypatch):
    lists = _lists(np.random.default_rng(0))
    path = str(tmp_path / "pickles") + "/"
    import_midi.save_pickle_cache(path, lists)
    assert sorted(p.name for p in (tmp_path / "pickles").iterdir()) == sorted(n + ".pickle" for n in import_midi.NAMES)
    back = import_midi.load_pickle_cache(path)
    for a, b in zip(lists, back):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    monkeypatch.setattr(settings, "load_from_pickle_instead_of_midi", True)
    monkeypatch.setattr(settings, "pickle_load_path", path)
    got = import_midi.import_midi_from_folder("data/original/")
    assert len(got) == 16 and np.array_equal(got[10][1], lists[10][1])          # X_train[1]
    train, test = vae_training.songs_from_pickle_cache(path, vars(settings))
    assert len(train) == 3 and len(test) == 2 and train[0]["X"].shape == lists[10][0].shape and train[1]["C"] == 1


def test_midi_parsing_is_refused_not_faked(monkeypatch):
    monkeypatch.setattr(settings, "load_from_pickle_instead_of_midi", False)
    with pytest.raises(NotImplementedError):
        import_midi.import_midi_from_folder("data/original/")
