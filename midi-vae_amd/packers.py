"""Host-side packers and the argmax decode of the MIDI-VAE hot path.

These build the ordered input / target / sample-weight lists the three models consume and turn
decoder outputs back into rolls.  Same names, argument meaning and output order as the reference's
module-level helpers (reference vae_definition.py:770-808, 816-865, 880-1045, 1048-1235), but written
as vectorised NumPy (the reference walks every (sample, step) in Python) and parameterised by an
explicit settings mapping instead of star-imported globals.  Results are pinned bit-exactly by
tests/golden/packers_decode.npz (captured from the reference's own functions).

Every function takes ``s`` = a mapping with the reference's settings names (see config.py).
"""
from __future__ import annotations

import numpy as np


def _g(s, name):
    return s[name] if isinstance(s, dict) else getattr(s, name)


def to_categorical(y, num_classes=None):
    """keras.utils.to_categorical semantics (used at reference vae_definition.py:919)."""
    y = np.array(y, dtype="int").ravel()
    if not num_classes:
        num_classes = int(np.max(y)) + 1
    out = np.zeros((y.shape[0], num_classes))
    out[np.arange(y.shape[0]), y] = 1
    return out


def _duration_categorical(D, T):
    """D (n,T) {0,1 or any nonzero = held} -> (n,T,2) one-hot [not-held, held]; only the first
    ``output_length`` steps are filled, like the reference loop (vae_definition.py:774-781)."""
    D = np.asarray(D)
    D_cat = np.zeros((D.shape[0], D.shape[1], 2))
    held = D[:, :T] != 0
    D_cat[:, :T, 0] = ~held
    D_cat[:, :T, 1] = held
    return D_cat


def _velocity_column(V, D_cat, s):
    V = np.copy(V)[..., None]
    if _g(s, "combine_velocity_and_held_notes"):
        T = _g(s, "output_length")
        held = D_cat[:, :T, 1] == 1
        assert np.all(V[:, :T, 0][held] == 0)  # a held step carries no hit velocity (:790,901)
        V[:, :T, 0][held] = 1
    return V


def prepare_encoder_input_list(s, X, I, V, D):
    """reference vae_definition.py:770-808.  Returns ``[X, I_tiled, V[...,None], D_cat]`` filtered by the
    meta_* switches, or bare ``X`` when no meta input is on."""
    n = X.shape[0]
    D_cat = _duration_categorical(D, _g(s, "output_length"))
    Vc = _velocity_column(V, D_cat, s)
    I_t = np.tile(np.expand_dims(I, axis=0), (n, 1, 1))
    mi, mv, mh = _g(s, "meta_instrument"), _g(s, "meta_velocity"), _g(s, "meta_held_notes")
    if not (mi or mv or mh):
        return X
    out = [X]
    if mi:
        out.append(I_t)
    if mv:
        out.append(Vc)
    if mh:
        out.append(D_cat)
    return out


def _additional_decoder_input(s, C_rows, S):
    """Shared by both packers (reference vae_definition.py:835-847, 968-980)."""
    rows = []
    if _g(s, "decoder_input_composer"):
        rows.extend(C_rows)
    if _g(s, "append_signature_vector_to_latent"):
        if len(rows) > 0:
            rows = np.append(np.asarray(rows), S, axis=1)
        else:
            rows.extend(S)
    return np.asarray(rows)


def prepare_decoder_input(s, R, C, S, H=None):
    """reference vae_definition.py:816-865.  ``[start, R, (gt), H, (additional), instr_start, vel_start, ...]``.
    With ``H is None`` the history is R rolled by one window (zeros first)."""
    n = R.shape[0]
    out = [np.zeros((n, _g(s, "output_dim"))), R]
    if _g(s, "teacher_force"):
        out.append(np.zeros((n, _g(s, "input_length"), _g(s, "output_dim"))))
    if _g(s, "history"):
        if H is not None:
            out.append(H)
        else:
            hist = np.zeros(R.shape)
            hist[1:] = R[:-1]
            out.append(hist)
    if _g(s, "decoder_additional_input"):
        out.append(_additional_decoder_input(s, C, S))
    if _g(s, "meta_instrument"):
        out.append(np.zeros((n, _g(s, "meta_instrument_dim"))))
    if _g(s, "meta_velocity"):
        out.append(np.zeros((n,)))
    if _g(s, "meta_held_notes"):
        out.append(np.zeros((n, 2)))
    if _g(s, "meta_next_notes"):
        out.append(np.zeros((n, _g(s, "output_dim"))))
    return out


def prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=False):
    """reference vae_definition.py:880-1045.  Input order ``[X, Y_start, (Y), H, (additional), instr_start, I,
    vel_start, V, held_start, D, next_start]``; targets ``[Y, I, V, D, N, C, S, C, C]``; sample weights
    ``[w_notes (n,T), w_style, w_signature, w_comp_notes, w_comp_instr, w_instr, w_vel, w_held, w_next]``
    (that order - weights of the classifier heads precede the meta heads, reference :928-961 vs :982-1028)."""
    n = X.shape[0]
    T = _g(s, "output_length")
    D_cat = _duration_categorical(D, T)
    Vc = _velocity_column(V, D_cat, s)
    Dm = D_cat
    N = None
    if _g(s, "meta_next_notes"):
        N = Y[1:]
        X, Y, Vc, Dm, S, H = X[:-1], Y[:-1], Vc[:-1], Dm[:-1], S[:-1], H[:-1]
        n = X.shape[0]
    Y_start = np.zeros((n, Y.shape[2]))
    C_rows = np.asarray([to_categorical(C, num_classes=_g(s, "num_classes"))] * n).squeeze()
    I_t = np.tile(np.expand_dims(I, axis=0), (n, 1, 1))

    x_list = [X, Y_start]
    y_list = [Y]
    w_list = None
    if return_sample_weight:
        w_notes = np.ones((n, T))
        if _g(s, "include_silent_note") and _g(s, "silent_weight") != 1.0:     # (weight 1.0 - the default - leaves the ones as they
            w_notes[Y[:, :, -1] == 1] = _g(s, "silent_weight")                 #  are: no strided pass over the window tensor)
        w_list = [w_notes]
        for flag in ("include_composer_decoder", "signature_decoder", "composer_decoder_at_notes_output",
                     "composer_decoder_at_instrument_output"):
            if _g(s, flag):
                w_list.append(np.ones((n,)))
    if _g(s, "teacher_force"):
        x_list.append(Y)
    if _g(s, "history"):
        x_list.append(H)
    if _g(s, "decoder_additional_input"):
        x_list.append(_additional_decoder_input(s, C_rows, S))
    if _g(s, "meta_instrument"):
        x_list += [np.zeros((n, _g(s, "meta_instrument_dim"))), I_t]
        y_list.append(I_t)
        if return_sample_weight:
            w_list.append(np.ones((n,)))
    if _g(s, "meta_velocity"):
        x_list += [np.zeros((n,)), Vc]
        y_list.append(Vc)
        if return_sample_weight:
            w_list.append(np.ones((n,)))
    if _g(s, "meta_held_notes"):
        x_list += [np.zeros((n, 2)), Dm]
        y_list.append(Dm)
        if return_sample_weight:
            w_list.append(np.ones((n,)))
    if _g(s, "meta_next_notes"):
        x_list.append(np.zeros((n, _g(s, "output_dim"))))
        y_list.append(N)
        if return_sample_weight:
            w_list.append(np.ones((n,)))
    if _g(s, "include_composer_decoder"):
        y_list.append(C_rows)
    if _g(s, "signature_decoder"):
        y_list.append(S)
    if _g(s, "composer_decoder_at_notes_output"):
        y_list.append(C_rows)
    if _g(s, "composer_decoder_at_instrument_output"):
        y_list.append(C_rows)
    if return_sample_weight:
        # the reference returns a bare array when only the notes weight exists (:930-940)
        return x_list, y_list, (w_list if len(w_list) > 1 else w_list[0])
    return x_list, y_list


# ------------------------------------------------------------------------------------------------
# decode (argmax path is the bit-exact contract; 'choice' follows the same formula with NumPy's RNG)
# ------------------------------------------------------------------------------------------------

def argmax_index_rows(P):
    """Row-wise note index of reference ``sample_vector(v, 'argmax')`` (vae_definition.py:1048-1067):
    first maximum; rows whose sum is not > 0 give index 0.  P: (..., K) -> int64 (...)."""
    P = np.asarray(P)
    idx = np.argmax(P, axis=-1)
    idx = np.where(np.sum(P, axis=-1) > 0, idx, 0)
    return idx.astype(np.int64)


def sample_vector(s, vector, sample_method):
    """reference vae_definition.py:1048-1067."""
    if np.sum(vector) > 0:
        if sample_method == "argmax":
            return int(np.argmax(vector))
        if sample_method == "choice":
            p = vector / (np.sum(vector) * 1.0)
            p = np.log(p) / _g(s, "temperature")
            p = np.exp(p) / np.sum(np.exp(p))
            k = 0
            for _ in range(_g(s, "number_of_tries")):
                k = int(np.random.choice(len(p), p=p))
                if p[k] > _g(s, "cutoff_sample_threshold"):
                    break
            return k
        raise ValueError("unknown sample_method %r" % (sample_method,))
    return 0


def _choice_rows(s, P):
    flat = P.reshape(-1, P.shape[-1])
    return np.array([sample_vector(s, row, "choice") for row in flat], dtype=np.int64).reshape(P.shape[:-1])


def note_indices(s, Y, sample_method):
    Y = np.asarray(Y)
    assert Y.ndim in (2, 3)
    flat = Y.reshape(-1, Y.shape[-1])
    return argmax_index_rows(flat) if sample_method == "argmax" else _choice_rows(s, flat)


def notes_from_indices(s, idx, width):
    """index roll -> one-hot pitch roll; the trailing class is the silent note and leaves the row
    all-zero (reference vae_definition.py:1084-1093)."""
    idx = np.asarray(idx).ravel()
    out = np.zeros((idx.shape[0], _g(s, "high_crop") - _g(s, "low_crop")))
    keep = np.ones(idx.shape, bool)
    if _g(s, "include_silent_note"):
        keep = idx != width - 1
    rows = np.nonzero(keep)[0]
    out[rows, idx[rows]] = 1
    return out


def sample_notes_prediction(s, Y, sample_method):
    """reference vae_definition.py:1071-1095: (n,T,K) or (T,K) probabilities -> (n*T, high_crop-low_crop)."""
    Y = np.asarray(Y)
    return notes_from_indices(s, note_indices(s, Y, sample_method), Y.shape[-1])


def sample_instrument_prediction(s, I, sample_method):
    """reference vae_definition.py:1097-1107: one-hot of the sampled index, same shape as ``I``."""
    I = np.asarray(I)
    idx = argmax_index_rows(I) if sample_method == "argmax" else _choice_rows(s, I)
    out = np.zeros(I.shape)
    np.put_along_axis(out, idx[..., None], 1, axis=-1)
    return out


def sample_held_notes_prediction(s, D, sample_method):
    """reference vae_definition.py:1109-1122: flat int array of sampled classes."""
    D = np.asarray(D)
    idx = argmax_index_rows(D) if sample_method == "argmax" else _choice_rows(s, D)
    return int(idx) if D.ndim == 1 else np.asarray(idx.ravel(), dtype=int)


def apply_velocity_rules(s, Y, V):
    """Velocity post-rules of reference vae_definition.py:1156-1190 on flat rolls (Y one-hot w/o silent column,
    V flat predicted velocities): silent steps get 0; per voice, a sounding pitch change whose predicted
    velocity is below the played-note threshold inherits the previous played velocity."""
    V = V.copy()
    silent = np.sum(Y, axis=1) == 0
    V[silent] = 0
    if not _g(s, "override_sampled_pitches_based_on_velocity_info"):
        return V
    thr = _g(s, "velocity_threshold_such_that_it_is_a_played_note")
    mv = _g(s, "max_voices")
    pitch = np.where(silent, -1, np.argmax(Y, axis=1))
    for voice in range(mv):
        p_roll = pitch[voice::mv]
        v_roll = V[voice::mv]          # a view: reads see this voice's earlier writes, like the reference
        prev_pitch, prev_vel = -1, 0.0
        for i in range(p_roll.shape[0]):
            p, vel = p_roll[i], v_roll[i]
            vel_silent = vel < thr
            if vel_silent:
                if p != -1 and prev_pitch > 0 and prev_pitch != p:
                    V[i * mv + voice] = prev_vel
            elif p == -1:
                V[i * mv + voice] = 0
            prev_pitch = p
            if not vel_silent:
                prev_vel = vel
    return V


def process_decoder_outputs(s, decoder_outputs, sample_method):
    """reference vae_definition.py:1131-1225 -> (Y, I, V, D, N).  Reproduces the reference's head indexing:
    with a list input, element 1 is decoded as the instrument head whenever any meta head is on."""
    T = _g(s, "output_length")
    Y = I = V = D = N = None
    if isinstance(decoder_outputs, (list, tuple)):
        Y = sample_notes_prediction(s, decoder_outputs[0], sample_method)
        count = 1
        if (_g(s, "meta_instrument") or _g(s, "meta_velocity") or _g(s, "meta_held_notes")
                or _g(s, "meta_next_notes")):
            I = sample_instrument_prediction(s, decoder_outputs[count], sample_method)
            count += 1
        if _g(s, "meta_velocity"):
            V = apply_velocity_rules(s, Y, np.asarray(decoder_outputs[count])[:, :, 0].reshape(-1)[:Y.shape[0]])
            count += 1
        if _g(s, "meta_held_notes"):
            D = sample_held_notes_prediction(s, decoder_outputs[count], sample_method)
            count += 1
        if _g(s, "meta_next_notes"):
            N = sample_notes_prediction(s, decoder_outputs[count], sample_method)
            count += 1
    else:
        Y = sample_notes_prediction(s, decoder_outputs, sample_method)
    L = Y.shape[0]
    thr = _g(s, "velocity_threshold_such_that_it_is_a_played_note")
    if I is None:
        I = np.zeros((L // T, _g(s, "max_voices"), _g(s, "meta_instrument_dim")))
        I[:, 0] = 1
    if V is None:
        V = np.ones((L,)) * (thr + (1.0 - thr) * 0.5)
    if D is None:
        D = np.ones((L,))
        if _g(s, "meta_velocity"):
            D[V > thr] = 0
    if N is None:
        N = np.zeros(Y.shape)
    return Y, I, V, D, N


def process_autoencoder_outputs(s, autoencoder_outputs, sample_method):
    """reference vae_definition.py:1234-1235."""
    return process_decoder_outputs(s, autoencoder_outputs, sample_method)


# ------------------------------------------------------------------------------------------------
# small data-format helpers on either side of the path
# ------------------------------------------------------------------------------------------------

def programs_to_instrument_matrix(programs, instrument_attach_method, max_voices):
    """GM program numbers -> per-voice instrument feature rows (reference midi_functions.py:14-54)."""
    progs = np.asarray(list(programs), dtype=np.int64)
    rows = np.arange(len(progs))
    if instrument_attach_method == "1hot-instrument":
        M = np.zeros((max_voices, 128))
        M[rows, progs] = 1
    elif instrument_attach_method == "1hot-category":
        M = np.zeros((max_voices, 16))
        M[rows, progs // 8] = 1
    elif instrument_attach_method == "khot-instrument":
        # NB the reference sets a bit where the binary digit is ZERO (midi_functions.py:33-35)
        M = np.zeros((max_voices, 7))
        M[: len(progs)] = ((progs[:, None] >> np.arange(7)[None, :]) & 1) == 0
    elif instrument_attach_method == "khot-category":
        M = np.zeros((max_voices, 4))
        M[: len(progs)] = (((progs // 8)[:, None] >> np.arange(4)[None, :]) & 1) == 1
    else:
        raise ValueError("instrument_attach_method %r not implemented" % (instrument_attach_method,))
    return M


def monophonic_to_khot_pianoroll(pianoroll, max_voices, set_all_nonzero_to_1=True):
    """Fold the interleaved monophonic roll back to a k-hot roll per tick (reference data_class.py:241-252)."""
    assert max_voices > 1
    n = pianoroll.shape[0] // max_voices
    out = np.zeros((n, pianoroll.shape[1]))
    np.add.at(out, np.arange(pianoroll.shape[0]) // max_voices, pianoroll)
    if set_all_nonzero_to_1:
        out[np.nonzero(out)] = 1
    return out


def onehot_to_index(X):
    """(n,T,K) one-hot float roll -> uint8 index roll (the device-side packing of the window tensor)."""
    X = np.asarray(X)
    assert X.shape[-1] <= 256
    return np.argmax(X, axis=-1).astype(np.uint8)
