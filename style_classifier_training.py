#!/usr/bin/env python3
"""Training entrypoint with the shape of the reference's three classifier scripts (pitch_classifier.py, velocity_classifier.py,
instrument_classifier.py: model :89-103, per-song fit loop :223-245, test() with the confusion matrix :117-160) on the MI355X
engine.  The reference reads MIDI folders (import_midi.py, out of scope); songs here are synthetic piano-roll windows whose
statistics depend on the class, so the classifiers have something to learn.

    python style_classifier_training.py --kind pitch|velocity|instrument [--epochs 5] [--songs 12]
"""
import argparse
import time

import numpy as np

import settings
from midi_vae_amd.classifier import StyleClassifier
from midi_vae_amd.synth import make_windows, to_reference_format


def synthetic_songs(n_songs, s, seed, kind):
    rng = np.random.default_rng(seed)
    songs = []
    for i in range(n_songs):
        c = i % s["num_classes"]
        n_win = int(rng.integers(8, 48))
        # class-dependent statistics: class 1 plays sparser (more silent rows), hits softer and prefers the upper GM categories
        w = make_windows(n_win, s["output_length"], s["output_dim"], s["max_voices"], s["meta_instrument_dim"], s["num_classes"],
                         s["latent_dim"], seed=seed * 1000 + i, p_silent=0.25 + 0.25 * c)
        X, Y, _, I, V, D = to_reference_format(w, s["output_dim"], s["meta_instrument_dim"])
        V = V * (1.0 - 0.3 * c)
        if c:
            I = np.roll(I, 8, axis=1)
        songs.append(dict(X=X, V=V, I=I, C=c))
    return songs


def sample(song, kind, num_classes):
    """(inputs, targets) of one song, built as the reference scripts build them"""
    if kind == "pitch":
        X = song["X"]                                                        # pitch_classifier.py:225
    elif kind == "velocity":
        X = np.expand_dims(song["V"], 2)                                     # velocity_classifier.py:259-260
    else:
        X = np.expand_dims(song["I"], 0)                                     # instrument_classifier.py:231-232: ONE sample per song
    onehot = np.eye(num_classes)[song["C"]]
    Y = onehot if kind == "instrument" else np.asarray([onehot] * X.shape[0]).squeeze()
    return X, Y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="pitch", choices=["pitch", "velocity", "instrument"])
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--songs", type=int, default=12)
    ap.add_argument("--test-songs", type=int, default=4)
    ap.add_argument("--batch-size", type=int, default=512)                  # the scripts' batch_size
    ap.add_argument("--learning-rate", type=float, default=2e-5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()
    s = vars(settings)
    nc = s["num_classes"]
    input_dim = {"pitch": s["input_dim"], "velocity": 1, "instrument": s["meta_instrument_dim"]}[args.kind]
    model = StyleClassifier(args.kind, input_dim=input_dim, num_classes=nc, lstm_size=256, num_layers=2,
                            learning_rate=args.learning_rate, optimizer="Adam", compute_dtype=args.dtype)
    print(model.summary())
    train = synthetic_songs(args.songs, s, 1, args.kind)
    test = synthetic_songs(args.test_songs, s, 99, args.kind)
    for e in range(1, args.epochs + 1):
        t0 = time.time()
        order = np.random.permutation(len(train))
        loss = acc = 0.0
        for i in order:
            X, Y = sample(train[i], args.kind, nc)
            if X.shape[0] > 1 or args.kind == "instrument":
                hist = model.fit(X, Y, epochs=1, batch_size=args.batch_size, shuffle=False, verbose=False)
                model.reset_states()
                loss += np.mean(hist.history["loss"])
                acc += np.mean(hist.history["acc"])
        conf = np.zeros((nc, nc))
        tl = 0.0
        for sg in test:                                                      # test(), pitch_classifier.py:117-160
            X, Y = sample(sg, args.kind, nc)
            tl += model.evaluate(X, Y, batch_size=args.batch_size, verbose=False)[0]
            Yp = model.predict(X, batch_size=args.batch_size, verbose=False)
            for yv, yp in zip(np.atleast_2d(Y), Yp):
                conf[np.argmax(yp), np.argmax(yv)] += 1
        print("Epoch %d: train loss %.4f acc %.3f | test loss %.4f acc %.3f | %.2f s" % (
            e, loss / len(train), acc / len(train), tl / len(test), np.trace(conf) / conf.sum(), time.time() - t0))


if __name__ == "__main__":
    main()
