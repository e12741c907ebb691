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


def _loop_windows(Y, V, D, s):
    """what the reference's load_rolls produces for one song (import_midi.py:256-265, :296-345), row by row in plain Python: a silent
    column that is set where a row is empty; X = every max_voices-th row under song_completion; rows appended up to whole windows,
    each a silent note with velocity 0 and not held - and, as written there, the silent bit of EVERY row when no row had to be
    appended (the slice [-0:] is the whole array)"""
    def windows(rows, length, make_pad, quirk):
        rows = [list(r) if hasattr(r, "__len__") else r for r in rows]
        missing = (-len(rows)) % length
        if missing == 0 and quirk:
            for r in rows:
                r[-1] = 1.0
        rows = rows + [make_pad() for _ in range(missing)]
        return np.array([rows[i:i + length] for i in range(0, len(rows), length)], dtype=float)

    width = len(Y[0])
    notes = []
    for row in np.asarray(Y, dtype=float):
        row = list(row)
        if s["include_silent_note"]:
            row.append(0.0 if sum(row) else 1.0)
        notes.append(row)
    full = width + (1 if s["include_silent_note"] else 0)
    silent_row = lambda: [0.0] * (full - 1) + [1.0] if s["include_silent_note"] else [0.0] * full
    inputs = notes[::s["max_voices"]] if s["song_completion"] else notes
    X = windows(inputs, s["input_length"], silent_row, s["include_silent_note"])
    Yw = windows(notes, s["output_length"], silent_row, s["include_silent_note"])
    Vw = windows(list(np.asarray(V, dtype=float)), s["output_length"], lambda: 0.0, False)
    Dw = windows(list(np.asarray(D, dtype=float)), s["output_length"], lambda: 0.0, False)
    return X, Yw, Vw, Dw


def _unrolled_song(rng, steps, width=60):
    Y = np.zeros((steps, width))
    sounding = rng.random(steps) < 0.7
    Y[np.nonzero(sounding)[0], rng.integers(0, width, int(sounding.sum()))] = 1
    return Y, rng.random(steps) * sounding, (rng.random(steps) < 0.3).astype(float)


@pytest.mark.parametrize("steps", [1, 37, 63, 65, 200])
@pytest.mark.parametrize("completion", [False, True])
def test_windows_from_unrolled_rolls_equal_the_references_statements(steps, completion):
    s = dict(include_silent_note=True, max_voices=4, song_completion=completion, input_length=16 if completion else 64,
             output_length=64, high_crop=84, low_crop=24)
    Y, V, D = _unrolled_song(np.random.default_rng(steps), steps)
    want = _loop_windows(Y, V, D, s)
    for a, b in zip(import_midi.windows_from_unrolled_rolls(Y, V, D, s, as_written=True), want):
        assert a.shape == b.shape and np.array_equal(a, b)
    got = import_midi.windows_from_unrolled_rolls(Y, V, D, s)
    whole = [((steps + 3) // 4 if completion else steps) % s["input_length"] == 0, steps % 64 == 0, False, False]   # (the next test)
    for a, b, w in zip(got, want, whole):
        assert a.shape == b.shape and (w or np.array_equal(a, b))
    assert np.all(got[0].sum(-1) == 1) and np.all(got[1].sum(-1) == 1)          # one-hot rows: what the engine's staging takes
    assert got[0].shape[-1] == 61 and got[1].shape[1:] == (64, 61) and got[2].shape == got[1].shape[:2] == got[3].shape


def test_a_song_of_whole_windows_gets_every_silent_bit_in_the_reference_and_not_here_by_default():
    """``X[-padding_length:, -1] = 1`` with padding_length 0 is ``X[0:, -1] = 1`` (reference import_midi.py:313-314, 327-328)"""
    s = dict(include_silent_note=True, max_voices=4, song_completion=False, input_length=64, output_length=64, high_crop=84, low_crop=24)
    Y, V, D = _unrolled_song(np.random.default_rng(1), 128)
    want = _loop_windows(Y, V, D, s)
    assert np.all(want[0][..., -1] == 1) and np.any(want[0].sum(-1) == 2)       # the quirk: two-hot rows
    got = import_midi.windows_from_unrolled_rolls(Y, V, D, s, as_written=True)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    sane = import_midi.windows_from_unrolled_rolls(Y, V, D, s)
    assert np.all(sane[0].sum(-1) == 1) and np.array_equal(sane[0][..., :60].reshape(-1, 60), Y)
    assert np.array_equal(sane[2], want[2]) and np.array_equal(sane[3], want[3])


def test_a_decoded_song_goes_back_into_windows_the_staging_accepts():
    """process_decoder_outputs leaves silent steps as all-zero rows without the silent column (vae_definition.py:1084-1093): fed to the
    one-hot validator as they are they are refused with a pointer to the way back; through the windowing they are valid input"""
    from midi_vae_amd import packers, staging
    from midi_vae_amd.config import build_settings
    s = build_settings(input_length=4, output_length=4)                           # (x max_voices = 16 rows per window)
    rng = np.random.default_rng(2)
    P = rng.random((3, s["output_length"], s["output_dim"]))
    P[:, ::3, -1] = 5.0                                                           # every third step: silence wins
    song = packers.sample_notes_prediction(s, P, "argmax")
    assert song.shape == (3 * s["output_length"], 60) and np.any(song.sum(1) == 0)
    with pytest.raises(NotImplementedError, match="windows_from_unrolled_rolls"):
        staging.host_onehot_to_index(song.reshape(3, s["output_length"], 60))
    X, Y, V, D = import_midi.windows_from_unrolled_rolls(song, np.zeros(len(song)), np.zeros(len(song)), s)
    idx = staging.host_onehot_to_index(X)
    assert idx.shape == (3, s["input_length"]) and np.array_equal(idx, packers.note_indices(s, P, "argmax").reshape(3, -1))
