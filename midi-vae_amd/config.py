"""Hyper-parameter surface of the MIDI-VAE hot path.

The reference keeps ~120 module-level globals in ``settings.py`` and star-imports them
everywhere (reference settings.py:1-245; vae_definition.py:12).  Here the same NAMES live
in one table of base knobs plus one derivation function, so a run can be re-parameterised
(``build_settings(cell_type='LSTM', latent_dim=64, ...)``) without editing a module, and the
repo-root ``settings.py`` simply publishes ``build_settings()`` as a flat namespace.

Differences from the reference, on purpose:
  * importing has NO filesystem side effect (reference settings.py:58-61 creates
    ``pickles/<unix-time>/`` at import); ``pickle_store_folder`` is still published.
  * the 128-entry General-MIDI instrument-name table (reference settings.py:252-397) is not
    carried: it only labels plots in the evaluation script, which is out of scope.
"""
from __future__ import annotations

import time
from types import SimpleNamespace

# name -> default.  Order is irrelevant; derived names are computed in ``_derive``.
_BASE = dict(
    # -- locations / generation (reference settings.py:8-32) --
    source_folder="data/original/", pickle_load_path="pickles/",
    temperature=1.0, sample_method="choice", cutoff_sample_threshold=0.0, number_of_tries=1,
    velocity_threshold_such_that_it_is_a_played_note=0.5,
    override_sampled_pitches_based_on_velocity_info=True, do_not_sample_in_evaluation=True,
    # -- classes / import (settings.py:36-102) --
    classes=("style1", "style2"), include_unknown=False, only_unknown=False, test_train_set=False,
    load_from_pickle_instead_of_midi=False, save_imported_midi_as_pickle=True, save_anything=True,
    split_equally_to_train_and_test=True, test_fraction=0.1, save_preprocessed_midi=False,
    smaller_training_set_factor=1.0, high_crop=84, low_crop=24, num_notes=128, SMALLEST_NOTE=16,
    MAXIMAL_NUMBER_OF_VOICES_PER_TRACK=1, MAX_VELOCITY=127.0, max_songs=100000, equal_mini_songs=False,
    attach_instruments=False, include_only_monophonic_instruments=False, max_voices=4,
    instrument_attach_method="1hot-category", song_completion=False,
    # -- VAE (settings.py:108-245).  *_length are PER-VOICE here and multiplied below. --
    input_length=16, output_length=16, lstm_size=256, latent_dim=256, batch_size=256,
    learning_rate=0.0002, beta=0.1, epsilon_std=0.01, save_step=10, shuffle_train_set=True,
    bidirectional=False, num_layers_encoder=2, num_layers_decoder=2, use_embedding=False, embedding_dim=0,
    decode=True, optimizer="Adam", vae_loss="categorical_crossentropy", activity_regularizer=None,
    reset_states=True, include_composer_feature=False, include_composer_decoder=True, composer_weight=0.1,
    split_lstm_vector=True, history=True, include_silent_note=True, activation="softmax", cell_type="GRU",
    silent_weight=1.0, teacher_force=False, epsilon_factor=0.0, extra_layer=True, lstm_activation="tanh",
    lstm_state_activation="tanh", decoder_input_composer=False, signature_vector_length=15,
    append_signature_vector_to_latent=False, meta_instrument=True, meta_instrument_activation="softmax",
    meta_instrument_weight=0.1, signature_decoder=False, signature_activation="tanh", signature_weight=1.0,
    composer_decoder_at_notes_output=False, composer_decoder_at_notes_weight=1.0,
    composer_decoder_at_notes_activation="softmax", composer_decoder_at_instrument_output=False,
    composer_decoder_at_instrument_weight=1.0, composer_decoder_at_instrument_activation="softmax",
    meta_velocity=True, meta_velocity_activation="sigmoid", meta_velocity_weight=1.0,
    meta_held_notes=False, meta_held_notes_activation="softmax", meta_held_notes_weight=0.1,
    combine_velocity_and_held_notes=False, meta_next_notes=False, meta_next_notes_weight=0.1,
    meta_next_notes_teacher_force=False, activation_before_splitting="tanh",
    epochs=2000, test_step=1, verbose=True, show_plot=False, save_plot=True,
    load_previous_checkpoint=False, previous_epoch=-1,
    previous_checkpoint_path=("models/autoencode/vae/1519767462-_inlen_256_outlen_256_beta_0.01_lr_0.0002_"
                              "lstmsize_256_latent_256_trainsize_859_testsize_99_shifted_True_epsstd_0.001/"),
    prior_mean=0.0, prior_std=1.0,
)

_INSTRUMENT_DIMS = {"1hot-category": 16, "khot-category": 4, "1hot-instrument": 128, "khot-instrument": 7}

# General-MIDI program // 8 category labels (standard GM grouping; reference settings.py:399-416)
_GM_CATEGORIES = ("piano", "chromatic percussion", "organs", "guitar", "bass", "strings", "ensemble", "brass",
                  "reed", "pipe", "synth lead", "synth pad", "synth effects", "ethnic", "percussive",
                  "sound effects")


def _derive(s: dict) -> dict:
    """Apply the reference's derivation order (settings.py:39-43,91-98,129-144,148-153,170-177,
    181-187,202-208,212,218,222-224,228)."""
    s["classes"] = list(s["classes"])
    s["num_classes"] = len(s["classes"]) + (1 if s["include_unknown"] else 0)
    s["new_num_notes"] = s["high_crop"] - s["low_crop"]
    instrument_dim = _INSTRUMENT_DIMS.get(s["instrument_attach_method"], 0)
    s["composer_length"] = s["num_classes"] if s["include_composer_feature"] else 0
    # per-voice lengths become interleaved-roll lengths (settings.py:140-144)
    s["output_length"] = s["output_length"] * s["max_voices"]
    if not s["song_completion"]:
        s["input_length"] = s["input_length"] * s["max_voices"]
    else:
        s["max_voices"] = 1
    if s["use_embedding"]:
        assert s["include_silent_note"]
    s["silent_dim"] = 1 if s["include_silent_note"] else 0
    s["decoder_additional_input"] = False
    s["decoder_additional_input_dim"] = 0
    if s["decoder_input_composer"]:
        s["decoder_additional_input"] = True
        s["decoder_additional_input_dim"] += s["num_classes"]
    if s["append_signature_vector_to_latent"]:
        s["decoder_additional_input"] = True
        s["decoder_additional_input_dim"] += s["signature_vector_length"]
    s["meta_instrument_dim"] = instrument_dim
    s["meta_instrument_length"] = s["max_voices"]
    if not s["attach_instruments"]:
        instrument_dim = 0
    s["instrument_dim"] = instrument_dim
    s["signature_dim"] = s["signature_vector_length"]
    any_composer_head = (s["composer_decoder_at_notes_output"] or s["composer_decoder_at_instrument_output"]
                         or s["include_composer_decoder"])
    s["num_composers"] = s["num_classes"] if any_composer_head else 0
    s["input_dim"] = s["new_num_notes"] + s["composer_length"] + s["silent_dim"] + instrument_dim
    s["output_dim"] = s["new_num_notes"] + s["silent_dim"] + instrument_dim
    s["meta_velocity_length"] = s["output_length"]
    s["meta_held_notes_length"] = s["output_length"]
    if s["combine_velocity_and_held_notes"]:
        s["meta_held_notes"] = False
    s["meta_next_notes_output_length"] = s["output_length"]
    s["instrument_category_names"] = list(_GM_CATEGORIES)
    s["t"] = str(int(round(time.time())))
    s["pickle_store_folder"] = "pickles/" + s["t"] + "/" if s["save_imported_midi_as_pickle"] else None
    return s


def build_settings(**overrides) -> dict:
    """Return the full flat settings dict.  ``overrides`` name BASE knobs (``input_length`` /
    ``output_length`` are per-voice, exactly like the literals in reference settings.py:108-109)."""
    unknown = set(overrides) - set(_BASE)
    if unknown:
        raise KeyError("unknown settings knob(s): %s" % sorted(unknown))
    s = dict(_BASE)
    s.update(overrides)
    return _derive(s)


def settings_namespace(**overrides) -> SimpleNamespace:
    return SimpleNamespace(**build_settings(**overrides))


# keyword names of VAE.create (reference vae_definition.py:40-102) -> settings names, as wired by
# the reference's entry script (vae_training.py:47-109).
CREATE_FROM_SETTINGS = dict(
    input_dim="input_dim", output_dim="output_dim", use_embedding="use_embedding", embedding_dim="embedding_dim",
    input_length="input_length", output_length="output_length", latent_rep_size="latent_dim",
    vae_loss="vae_loss", optimizer="optimizer", activation="activation", lstm_activation="lstm_activation",
    lstm_state_activation="lstm_state_activation", epsilon_std="epsilon_std", epsilon_factor="epsilon_factor",
    include_composer_decoder="include_composer_decoder", num_composers="num_composers",
    composer_weight="composer_weight", lstm_size="lstm_size", cell_type="cell_type",
    num_layers_encoder="num_layers_encoder", num_layers_decoder="num_layers_decoder",
    bidirectional="bidirectional", decode="decode", teacher_force="teacher_force",
    learning_rate="learning_rate", split_lstm_vector="split_lstm_vector", history="history", beta="beta",
    prior_mean="prior_mean", prior_std="prior_std", decoder_additional_input="decoder_additional_input",
    decoder_additional_input_dim="decoder_additional_input_dim", extra_layer="extra_layer",
    meta_instrument="meta_instrument", meta_instrument_dim="meta_instrument_dim",
    meta_instrument_length="meta_instrument_length", meta_instrument_activation="meta_instrument_activation",
    meta_instrument_weight="meta_instrument_weight", signature_decoder="signature_decoder",
    signature_dim="signature_dim", signature_activation="signature_activation",
    signature_weight="signature_weight", composer_decoder_at_notes_output="composer_decoder_at_notes_output",
    composer_decoder_at_notes_weight="composer_decoder_at_notes_weight",
    composer_decoder_at_notes_activation="composer_decoder_at_notes_activation",
    composer_decoder_at_instrument_output="composer_decoder_at_instrument_output",
    composer_decoder_at_instrument_weight="composer_decoder_at_instrument_weight",
    composer_decoder_at_instrument_activation="composer_decoder_at_instrument_activation",
    meta_velocity="meta_velocity", meta_velocity_length="meta_velocity_length",
    meta_velocity_activation="meta_velocity_activation", meta_velocity_weight="meta_velocity_weight",
    meta_held_notes="meta_held_notes", meta_held_notes_length="meta_held_notes_length",
    meta_held_notes_activation="meta_held_notes_activation", meta_held_notes_weight="meta_held_notes_weight",
    meta_next_notes="meta_next_notes", meta_next_notes_output_length="meta_next_notes_output_length",
    meta_next_notes_weight="meta_next_notes_weight", meta_next_notes_teacher_force="meta_next_notes_teacher_force",
    activation_before_splitting="activation_before_splitting",
)


def create_kwargs(settings: dict) -> dict:
    """The 61 keyword arguments the reference's entry script passes to ``VAE.create``."""
    return {k: settings[v] for k, v in CREATE_FROM_SETTINGS.items()}
