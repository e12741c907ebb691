#!/usr/bin/env python3
"""Style transfer by latent swap, scored by a style classifier - the inference pipeline of the reference's evaluation script
(vae_evaluation.py:2180-2181 encode, :2471-2478 swap the style dimensions of z, :2481-2483 decode window by window + argmax,
:78-91 / :2585-2625 classify the result) with every stage on the MI355X engine and batched: the windows of a song are independent
given (z', history), so the decoder runs them as ONE batch (BASELINE configs[4]) and nothing but one byte per row leaves the chip.

Synthetic songs (the reference reads MIDI folders); the VAE and the pitch classifier are trained here for a few epochs first, so
the printed numbers only show the pipeline working - not the paper's results.

    python style_transfer_eval.py [--epochs 3] [--songs 8]
"""
import argparse
import time

import numpy as np

import settings
import vae_definition
from midi_vae_amd.classifier import StyleClassifier
from midi_vae_amd.config import create_kwargs
from style_classifier_training import synthetic_songs
from vae_definition import VAE


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--songs", type=int, default=8)
    args = ap.parse_args()
    s = vars(settings)
    nc, bs = s["num_classes"], s["batch_size"]
    songs = synthetic_songs(args.songs, s, 5, "pitch")
    for sg in songs:
        n = sg["X"].shape[0]
        sg.update(Y=sg["X"], D=np.zeros(sg["V"].shape), S=np.zeros((n, s["signature_vector_length"])))
    model = VAE().create(**create_kwargs(s))
    clf = StyleClassifier("pitch", input_dim=s["input_dim"], num_classes=nc, learning_rate=2e-4)
    for e in range(args.epochs):
        hs = []
        for sg in songs:
            H = np.zeros((sg["X"].shape[0], s["latent_dim"]))
            x, y, w = vae_definition.prepare_autoencoder_input_and_output_list(sg["X"], sg["Y"], sg["C"], sg["I"], sg["V"], sg["D"],
                                                                               sg["S"], H, return_sample_weight=True)
            hs.append(model.autoencoder.fit(x, y, epochs=1, batch_size=bs, shuffle=False, sample_weight=w, verbose=False))
            clf.fit(sg["X"], np.tile(np.eye(nc)[sg["C"]][None], (sg["X"].shape[0], 1)), epochs=1, batch_size=bs, shuffle=False)
        print("epoch %d: VAE loss %.4f" % (e, np.mean([h.history["loss"][0] for h in hs])))
    t0 = time.time()
    n_win = kept = switched = 0
    for sg in songs:
        X, I, V, D, C = sg["X"], sg["I"], sg["V"], sg["D"], sg["C"]
        n = X.shape[0]
        z = model.encoder.predict(vae_definition.prepare_encoder_input_list(X, I, V, D), batch_size=bs)      # :2180-2181
        target = (C + 1) % nc
        z2 = z.copy()
        z2[:, C], z2[:, target] = z[:, target], z[:, C]                                                     # :2471-2478
        for name, zz in (("kept", z), ("switched", z2)):
            dec_in = vae_definition.prepare_decoder_input(zz, C, sg["S"], None)    # history = previous window's (switched) z, :2481
            idx = model.decoder.predict_note_indices(dec_in, batch_size=max(bs, n))                         # :2482-2483, fused
            pred = np.argmax(clf.predict(idx, batch_size=bs), axis=1)
            if name == "kept":
                kept += int(np.sum(pred == C))
            else:
                switched += int(np.sum(pred == target))
        n_win += n
    dt = time.time() - t0
    print("%d windows: classified as their own style after autoencoding %.1f %%, as the TARGET style after the latent swap %.1f %% "
          "(encode + 2 x decode + 2 x classify: %.0f windows/s)" % (n_win, 100.0 * kept / n_win, 100.0 * switched / n_win, n_win / dt))


if __name__ == "__main__":
    main()
