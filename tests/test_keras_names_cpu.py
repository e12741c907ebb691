"""midi-vae_amd/keras_names.py: the reference's checkpoint layout (Keras layer names / creation order, recurrentshop cell weights;
vae_training.py:966-978, vae_definition.py:443-728) <-> this package's tensor names, on a synthetic model (SURVEY f-4: the
weight files are missing upstream and h5py from this image)."""
import numpy as np
import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd import keras_names as kn
from midi_vae_amd.layout import ModelSpec, ParamLayout, init_params
from oracle import vae_oracle as vo


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "SimpleRNN"])
def test_every_tensor_of_the_layout_has_a_place_and_the_map_roundtrips(cell):
    spec = ModelSpec(cell=cell, H=64, Z=16, Din=61, Dout=61, T=8, V=4, ID=16, C=2, Le=2, Ld=2)
    params = init_params(spec, 3)
    enc, dec = kn.to_keras(spec, params)
    pre = {"GRU": "gru_", "LSTM": "lstm_", "SimpleRNN": "rnn_"}[cell]
    assert list(enc) == [pre + "1", pre + "2", pre + "meta_instrument", pre + "meta_velocity", "extra_instrument_after_concat_layer",
                         "extra_layer", "z_mean", "z_log_var"]                     # (the names the reference gives, :448-507)
    assert [a.shape for a in enc[pre + "1"]] == [(61, spec.G * 64), (64, spec.G * 64), (spec.G * 64,)]
    # decoder: cells bottom -> top, output Dense, initial-state Denses - per head, notes / instrument / velocity (:519-643)
    n_init = 2 if cell == "LSTM" else 1
    assert len(dec) == (2 + 1 + 2 * n_init) + 2 * (1 + 1 + n_init)
    back = kn.from_keras(spec, enc, dec)
    assert set(back) == set(ParamLayout.build(spec).oracle_names()) == set(params)
    for k, v in params.items():
        np.testing.assert_array_equal(back[k], v, err_msg=k)


def test_recurrentshop_lstm_gate_order_is_permuted_into_this_packages():
    """recurrentshop.cells.LSTMCell computes, in ITS gate order [f | i | c | o] (recalled):
         f = hs(.), i = hs(.), c' = f c + i tanh(.), h' = hs(.)_o tanh(c')
    A direct restatement of that with [f|i|c|o] weights must equal the oracle's LSTM step ([i|f|g|o]) on the converted weights."""
    rng = np.random.default_rng(0)
    H, D, B = 8, 5, 3
    Wrs, Urs, brs = rng.standard_normal((D, 4 * H)), rng.standard_normal((H, 4 * H)) * 0.3, rng.standard_normal(4 * H) * 0.1
    x, h, c = rng.standard_normal((B, D)), rng.standard_normal((B, H)) * 0.5, rng.standard_normal((B, H)) * 0.5
    a = x @ Wrs + brs + h @ Urs
    f, i = vo.hard_sigmoid(a[:, :H]), vo.hard_sigmoid(a[:, H:2 * H])
    c1 = f * c + i * np.tanh(a[:, 2 * H:3 * H])
    h1 = vo.hard_sigmoid(a[:, 3 * H:]) * np.tanh(c1)
    W, U, b = kn.cell_from_recurrentshop("LSTM", [Wrs, brs, Urs], H)
    hs, cs, _ = vo.rnn_forward("LSTM", (x @ W + b)[None], U, h, c)
    np.testing.assert_allclose(hs[1], h1, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(cs[1], c1, rtol=1e-13, atol=1e-14)
    for got, want in zip(kn.cell_to_recurrentshop("LSTM", W, U, b, H), [Wrs, brs, Urs]):
        np.testing.assert_array_equal(got, want)


def test_recurrentshop_gru_keeps_two_recurrent_kernels():
    rng = np.random.default_rng(1)
    H, D = 8, 5
    W, b = rng.standard_normal((D, 3 * H)), rng.standard_normal(3 * H)
    Uzr, Uh = rng.standard_normal((H, 2 * H)), rng.standard_normal((H, H))
    W2, U, b2 = kn.cell_from_recurrentshop("GRU", [W, b, Uzr, Uh], H)
    np.testing.assert_array_equal(U, np.concatenate([Uzr, Uh], 1))
    assert W2 is W and b2 is b
    got = kn.cell_to_recurrentshop("GRU", W, U, b, H)
    assert [g.shape for g in got] == [(D, 3 * H), (3 * H,), (H, 2 * H), (H, H)]


def test_a_checkpoint_with_another_number_of_layers_is_refused():
    spec = ModelSpec(cell="GRU", H=64, Z=16, Din=61, Dout=61, T=8, V=4, ID=16, C=2, Le=2, Ld=2)
    enc, dec = kn.to_keras(spec, init_params(spec, 0))
    with pytest.raises(ValueError, match="weighted layers"):
        kn.from_keras(spec, enc, dec[:-1])
