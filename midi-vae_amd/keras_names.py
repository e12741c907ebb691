"""Keras layer names / weight order of the reference's checkpoints <-> this package's parameter names (SURVEY f-4).

The reference saves ``{autoencoder,encoder,decoder}Epoch<e>.pickle`` with ``Model.save_weights`` (Keras HDF5,
``vae_training.py:966-978``) and loads them back ``by_name=False`` (``:115-130``), i.e. BY LAYER ORDER.  The files themselves are
not in the reference tree (``.MISSING_LARGE_BLOBS``) and this image has no h5py, so what is built here is the part that needs
neither: given the weights as ``[(layer name, [arrays in Keras' per-layer order]), ...]`` - what ``h5py`` would read out of
``model_weights/<layer>/<weight_names>`` - produce the named tensors ``layout.ParamLayout`` / ``VAE.set_weights`` take, and back.

What is READ from the reference and what is RECALLED (as SURVEY Appendix A flags it):
  * encoder [R]: every layer is named explicitly - ``gru_<k>`` / ``lstm_<k>`` / ``rnn_<k>`` (k = 1..num_layers_encoder,
    ``vae_definition.py:448-461``), ``<cell>_meta_instrument`` / ``_meta_velocity`` / ``_meta_held_notes`` (``:465-479``),
    ``extra_instrument_after_concat_layer`` (``:484``), ``extra_layer`` (``:487``), ``z_mean`` / ``z_log_var`` (``:506-507``).
    A Keras recurrent layer holds [kernel (in, G*H), recurrent_kernel (H, G*H), bias (G*H)] with gates [z|r|h] (GRU) and
    [i|f|c|o] (LSTM) - this package's own order: no permutation [K].
  * decoder [R] for the ORDER of creation (``:519-728``: per head the cells bottom -> top, the output Dense, then one initial-state
    Dense per state and layer), [RS] for what a recurrentshop cell holds: kernel (in, G*H), bias (G*H), recurrent kernel(s) -
    GRUCell keeps the z / r recurrent kernel (H, 2H) and the candidate's (H, H) apart, LSTMCell orders its gates [f|i|c|o]
    (SURVEY A.5).  Its layers are unnamed (auto-numbered ``dense_<n>`` ...): they are matched by position and shape.
Nothing here can be checked against a real checkpoint; ``tests/test_keras_names_cpu.py`` pins the map on a synthetic model and
checks the recurrentshop gate permutation against a direct restatement of that cell's equations."""
from __future__ import annotations

import numpy as np

_PRE = {"GRU": "gru_", "LSTM": "lstm_", "SimpleRNN": "rnn_"}
_G = {"GRU": 3, "LSTM": 4, "SimpleRNN": 1}


def encoder_layers(spec):
    """[(Keras layer name, kind, this package's prefix)] in the reference's creation order; kind 'rnn' or 'dense'"""
    if getattr(spec, "bidirectional", False):
        raise NotImplementedError("bidirectional encoder: Bidirectional(...) wraps two copies per layer (forward_/backward_ sub-layers)")
    pre = _PRE[spec.cell]
    out = [(pre + str(l + 1), "rnn", "enc.notes.%d" % l) for l in range(spec.Le)]
    if spec.meta_instrument:
        out.append((pre + "meta_instrument", "rnn", "enc.instr"))
    if spec.meta_velocity:
        out.append((pre + "meta_velocity", "rnn", "enc.vel"))
    if getattr(spec, "meta_held", False):
        out.append((pre + "meta_held_notes", "rnn", "enc.held"))
    if spec.meta_instrument or spec.meta_velocity:          # (the reference's condition repeats meta_instrument, :483)
        out.append(("extra_instrument_after_concat_layer", "dense", "enc.pack"))
    if spec.extra_layer:
        out.append(("extra_layer", "dense", "enc.extra"))
    out += [("z_mean", "dense", "enc.zmean"), ("z_log_var", "dense", "enc.zlogvar")]
    return out


def decoder_layers(spec):
    """[(kind, prefix)] in the reference's creation order (vae_definition.py:519-728); kind 'cell' (recurrentshop) or 'dense'"""
    out = []

    def head(cells, out_dense, inits):
        out.extend(("cell", c) for c in cells)
        out.append(("dense", out_dense))
        out.extend(("dense", i) for i in inits)

    ns = spec.nstate
    head(["dec.notes.%d" % l for l in range(spec.Ld)], "dec.notes.out",
         ["dec.notes.init.%d.%d" % (l, s) for l in range(spec.Ld) for s in range(ns)])
    if spec.meta_instrument:
        head(["dec.instr.cell"], "dec.instr.out", ["dec.instr.init.%d" % s for s in range(ns)])
    if spec.meta_velocity:
        head(["dec.vel.cell"], "dec.vel.out", ["dec.vel.init.%d" % s for s in range(ns)])
    if getattr(spec, "meta_held", False):
        head(["dec.held.cell"], "dec.held.out", ["dec.held.init.%d" % s for s in range(ns)])
    if getattr(spec, "meta_next", False):
        head(["dec.next.%d" % l for l in range(spec.Ld)], "dec.next.out",
             ["dec.next.init.%d.%d" % (l, s) for l in range(spec.Ld) for s in range(ns)])
    return out


def _lstm_rs_perm(H):
    """column index of this package's [i|f|g|o] gate block inside recurrentshop's [f|i|c|o]"""
    blk = lambda k: np.arange(k * H, (k + 1) * H)
    return np.concatenate([blk(1), blk(0), blk(2), blk(3)])


def cell_from_recurrentshop(cell, arrays, H):
    """recurrentshop cell weights [kernel, bias, recurrent...] -> (W, U, b) in this package's gate order"""
    G = _G[cell]
    kernel, bias, rec = arrays[0], arrays[1], list(arrays[2:])
    if cell == "GRU":
        zr = next(a for a in rec if a.shape == (H, 2 * H))
        hh = next(a for a in rec if a.shape == (H, H))
        return kernel, np.concatenate([zr, hh], 1), bias
    U = rec[0]
    if cell == "LSTM":
        p = _lstm_rs_perm(H)
        return kernel[:, p], U[:, p], bias[p]
    assert U.shape == (H, G * H)
    return kernel, U, bias


def cell_to_recurrentshop(cell, W, U, b, H):
    if cell == "GRU":
        return [W, b, U[:, :2 * H], U[:, 2 * H:]]
    if cell == "LSTM":
        inv = np.argsort(_lstm_rs_perm(H))
        return [W[:, inv], b[inv], U[:, inv]]
    return [W, b, U]


def from_keras(spec, encoder_weights, decoder_weights):
    """``encoder_weights``: {Keras layer name: [arrays]}; ``decoder_weights``: [[arrays], ...] of the decoder's weighted layers in
    creation order.  Returns {this package's tensor name: array} (float32), every tensor of the layout exactly once."""
    out = {}
    for name, kind, prefix in encoder_layers(spec):
        arrs = encoder_weights[name]
        if kind == "rnn":
            out[prefix + ".W"], out[prefix + ".U"], out[prefix + ".b"] = arrs
        else:
            out[prefix + ".W"], out[prefix + ".b"] = arrs
    layers = decoder_layers(spec)
    if len(layers) != len(decoder_weights):
        raise ValueError("the decoder has %d weighted layers here, the checkpoint %d" % (len(layers), len(decoder_weights)))
    for (kind, prefix), arrs in zip(layers, decoder_weights):
        if kind == "cell":
            out[prefix + ".W"], out[prefix + ".U"], out[prefix + ".b"] = cell_from_recurrentshop(spec.cell, arrs, spec.H)
        else:
            out[prefix + ".W"], out[prefix + ".b"] = arrs
    return {k: np.asarray(v, np.float32) for k, v in out.items()}


def to_keras(spec, params):
    """the inverse of from_keras: ({layer name: [arrays]}, [[arrays], ...])"""
    enc = {}
    for name, kind, prefix in encoder_layers(spec):
        enc[name] = [params[prefix + ".W"], params[prefix + ".U"], params[prefix + ".b"]] if kind == "rnn" else \
            [params[prefix + ".W"], params[prefix + ".b"]]
    dec = []
    for kind, prefix in decoder_layers(spec):
        dec.append(cell_to_recurrentshop(spec.cell, params[prefix + ".W"], params[prefix + ".U"], params[prefix + ".b"], spec.H)
                   if kind == "cell" else [params[prefix + ".W"], params[prefix + ".b"]])
    return enc, dec
