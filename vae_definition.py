"""Drop-in for the reference's ``vae_definition`` module: ``VAE`` plus the module-level packers / decoders, bound
to the flat ``settings`` namespace at call time exactly like the reference's star-imported globals
(reference vae_definition.py:12, 770-1235).  Implementation lives in ``midi_vae_amd`` (model.py, packers.py)."""
import midi_vae_amd  # noqa: F401
import settings as _settings
from midi_vae_amd import packers as _pk
from midi_vae_amd.model import VAE  # noqa: F401


def _s():
    return vars(_settings)


def prepare_encoder_input_list(X, I, V, D):
    return _pk.prepare_encoder_input_list(_s(), X, I, V, D)


def prepare_decoder_input(R, C, S, H=None):
    return _pk.prepare_decoder_input(_s(), R, C, S, H)


def prepare_autoencoder_input_and_output_list(X, Y, C, I, V, D, S, H, return_sample_weight=False):
    return _pk.prepare_autoencoder_input_and_output_list(_s(), X, Y, C, I, V, D, S, H, return_sample_weight)


def sample_vector(vector, sample_method):
    return _pk.sample_vector(_s(), vector, sample_method)


def sample_notes_prediction(Y, sample_method):
    return _pk.sample_notes_prediction(_s(), Y, sample_method)


def sample_instrument_prediction(I, sample_method):
    return _pk.sample_instrument_prediction(_s(), I, sample_method)


def sample_held_notes_prediction(D, sample_method):
    return _pk.sample_held_notes_prediction(_s(), D, sample_method)


def process_decoder_outputs(decoder_outputs, sample_method):
    return _pk.process_decoder_outputs(_s(), decoder_outputs, sample_method)


def process_autoencoder_outputs(autoencoder_outputs, sample_method):
    return _pk.process_autoencoder_outputs(_s(), autoencoder_outputs, sample_method)
