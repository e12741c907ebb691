#!/usr/bin/env python3
"""Training entrypoint with the shape of the reference's ``vae_training.py`` (reference :47-130 model build,
:728-815 per-song fit loop with the history pre-pass, :243-351 test(), :817-961 metric bookkeeping with KL derived by
subtraction, :966-978 checkpoints) on the MI355X engine.

The reference imports MIDI files with pretty_midi (import_midi.py), which is outside this repo's scope: songs here are
synthetic piano-roll windows with the same tensor layout (midi_vae_amd/synth.py), or ``--pickle`` pointing at arrays
saved by a user's own importer (an ``.npz`` with X, C, I, V per song).  Plots / tikz / the ~30 per-epoch pickles of
the reference are not reproduced.

    python vae_training.py --epochs 3 --songs 6                         (single GPU)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 vae_training.py --epochs 3   (DP: every
        rank walks the same songs; each minibatch of <= batch_size windows is sharded over the ranks inside fit)
"""
import argparse
import os
import time

import numpy as np

import settings
import vae_definition
from midi_vae_amd.config import create_kwargs
from midi_vae_amd.synth import make_windows, to_reference_format
from vae_definition import VAE


def synthetic_songs(n_songs, s, seed):
    rng = np.random.default_rng(seed)
    songs = []
    for i in range(n_songs):
        n_win = int(rng.integers(max(2, s["batch_size"] // 4), 2 * s["batch_size"]))       # ragged: songs differ in length
        w = make_windows(n_win, s["output_length"], s["output_dim"], s["max_voices"], s["meta_instrument_dim"],
                         s["num_classes"], s["latent_dim"], seed=seed * 1000 + i)
        X, Y, _, I, V, D = to_reference_format(w, s["output_dim"], s["meta_instrument_dim"])
        songs.append(dict(X=X, Y=Y, C=i % s["num_classes"], I=I, V=V, D=D,
                          S=np.zeros((n_win, s["signature_vector_length"]))))
    return songs


def songs_from_pickle_cache(path, s):
    """(train, test) song dicts from the reference's dataset cache (import_midi.load_pickle_cache; reference vae_training.py:142)"""
    import import_midi
    (V_tr, V_te, D_tr, D_te, _, _, I_tr, I_te, Y_tr, Y_te, X_tr, X_te, c_tr, c_te, _, _) = import_midi.load_pickle_cache(path)

    def songs(X, Y, C, I, V, D):
        return [dict(X=np.asarray(x), Y=np.asarray(y), C=int(c), I=np.asarray(i), V=np.asarray(v), D=np.asarray(d),
                     S=np.zeros((np.asarray(x).shape[0], s["signature_vector_length"])))
                for x, y, c, i, v, d in zip(X, Y, C, I, V, D)]

    return songs(X_tr, Y_tr, c_tr, I_tr, V_tr, D_tr), songs(X_te, Y_te, c_te, I_te, V_te, D_te)


def history_for(model, song, s, use_encoder, on_device=True):
    """reference vae_training.py:788-798 - zeros in epoch 0, else the previous window's SAMPLED z (a fresh epsilon: one
    extra encoder forward per song in the reference).  ``on_device``: a deferred model.DeviceLatent - the draw is taken now, the
    encoder pass is fused into the fit / evaluate call that receives it (first minibatch: out of its own encoder forward; the
    rest: one chip-filling forward-only pass), z stays in HBM and the roll H[1:] = z[:-1], H[0] = 0 happens where the train
    step reads it; else the reference's host arrays."""
    n = song["X"].shape[0]
    if not (s["history"] and use_encoder):
        return np.zeros((n, s["latent_dim"]))
    enc_in = vae_definition.prepare_encoder_input_list(song["X"], song["I"], song["V"], song["D"])
    if on_device:
        return model.encoder.predict(enc_in, batch_size=s["batch_size"], verbose=False, device=True)
    z = model.encoder.predict(enc_in, batch_size=s["batch_size"], verbose=False)
    H = np.zeros(z.shape)
    H[1:] = z[:-1]
    return H


def run_epoch(model, songs, s, epoch, train):
    names = model.autoencoder.metrics_names
    enum, seen, total = [], {}, {n: names.count(n) for n in names}
    for n in names:                                            # reference vae_training.py:172-187
        seen[n] = seen.get(n, 0) + 1
        enum.append("%s_%d" % (n, seen[n]) if total[n] > 1 else n)
    agg, pending = {}, []
    for song in songs:
        H = history_for(model, song, s, use_encoder=(epoch > 0 or not train))
        if train:
            x, y, w = vae_definition.prepare_autoencoder_input_and_output_list(
                song["X"], song["Y"], song["C"], song["I"], song["V"], song["D"], song["S"], H, return_sample_weight=True)
            hist = model.autoencoder.fit(x, y, epochs=1, batch_size=s["batch_size"], shuffle=False, sample_weight=w,
                                         verbose=False)
            if s["reset_states"]:
                model.autoencoder.reset_states()
            pending.append(hist)        # (read after the last song: History is filled on first access - reading it here, as the
            continue                    #  reference does, would make the host wait for the device once per song)
        else:
            x, y = vae_definition.prepare_autoencoder_input_and_output_list(
                song["X"], song["Y"], song["C"], song["I"], song["V"], song["D"], song["S"], H)
            vals = dict(zip(enum, model.autoencoder.evaluate(x, y, batch_size=s["batch_size"], verbose=False)))
        for k, v in vals.items():
            agg[k] = agg.get(k, 0.0) + v
    for hist in pending:
        for k, v in hist.history.items():
            agg[k] = agg.get(k, 0.0) + float(np.mean(v))
    out = {k: v / max(len(songs), 1) for k, v in agg.items()}
    weighted = out.get("decoder_loss_1", out.get("decoder_loss", 0.0))
    if s["meta_instrument"]:
        weighted += s["meta_instrument_weight"] * out.get("decoder_loss_2", 0.0)
    if s["meta_velocity"]:
        weighted += s["meta_velocity_weight"] * out.get("decoder_loss_%d" % (2 + int(s["meta_instrument"])), 0.0)
    if s["include_composer_decoder"]:
        weighted += s["composer_weight"] * out.get("composer_decoder_loss", 0.0)
    out["kl_loss"] = (out["loss"] - weighted) / s["beta"]      # reference vae_training.py:946-959
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--songs", type=int, default=8)
    ap.add_argument("--test-songs", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model-path", default="models/autoencode/vae/")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend when WORLD_SIZE > 1")
    ap.add_argument("--pickle-dir", default="", help="a dataset cache written by the reference (import_midi.py:548-571: V_train.pickle ... "
                    "test_paths.pickle) instead of synthetic songs")
    args = ap.parse_args()
    s = vars(settings)

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dp = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}))
        from midi_vae_amd.dp import DataParallel
        dp = DataParallel(dist)

    print("creating model...")
    model = VAE()
    model.create(compute_dtype=args.dtype, device="cuda:%d" % local, **create_kwargs(s))
    # Data parallel: EVERY rank walks the same song list in the same order and calls fit with the same arguments; fit splits
    # each minibatch of <= batch_size consecutive windows over the ranks (dp.shard_bounds).  All ranks therefore run the same
    # number of optimizer steps and collectives by construction - ragged songs and ranks with empty shards included.
    model.set_data_parallel(dp)
    if rank == 0:
        print(model.autoencoder.summary())
    if s["load_previous_checkpoint"]:
        for view, name in ((model.autoencoder, "autoencoder"), (model.encoder, "encoder"), (model.decoder, "decoder")):
            view.load_weights(s["previous_checkpoint_path"] + name + "Epoch" + str(s["previous_epoch"]) + ".pickle", by_name=False)
    if args.pickle_dir:
        train, test = songs_from_pickle_cache(args.pickle_dir, s)
    else:
        train = synthetic_songs(args.songs, s, seed=1)         # the SAME songs on every rank (sharded inside fit)
        test = synthetic_songs(args.test_songs, s, seed=999)
    order_rng = np.random.default_rng(4321)                     # ... in the SAME order (the reference's shuffle is unseeded)
    path = os.path.join(args.model_path, "%s-_ls_inlen_%d_outlen_%d_beta_%s_lr_%s_lstmsize_%d_latent_%d" % (
        s["t"], s["input_length"], s["output_length"], s["beta"], s["learning_rate"], s["lstm_size"], s["latent_dim"]))
    start = s["previous_epoch"] if s["load_previous_checkpoint"] else 0
    for e in range(start, start + args.epochs):
        t0 = time.time()
        order = order_rng.permutation(len(train)) if s["shuffle_train_set"] else np.arange(len(train))
        tr = run_epoch(model, [train[i] for i in order], s, e, train=True)
        n_win = sum(sg["X"].shape[0] for sg in train)
        if rank == 0:
            print("Epoch %d: train loss %.4f notes %.4f acc %.4f kl %.5f | %.0f windows/s" % (
                e, tr["loss"], tr.get("decoder_loss_1", tr["loss"]), tr.get("decoder_acc_1", 0.0), tr["kl_loss"],
                n_win / (time.time() - t0)))
        if e % s["test_step"] == 0:
            # EVERY rank: encoder.predict and evaluate shard each song over the ranks (and draw from the same generator on every
            # rank, so the epsilon streams of the replicas stay identical) - reference vae_training.py:286-300 on all GPUs
            te = run_epoch(model, test, s, e, train=False)
            if rank == 0:
                print("         test  loss %.4f notes %.4f acc %.4f kl %.5f" % (
                    te["loss"], te.get("decoder_loss_1", te["loss"]), te.get("decoder_acc_1", 0.0), te["kl_loss"]))
        if e % s["save_step"] == 0 and s["save_anything"] and rank == 0:
            os.makedirs(path, exist_ok=True)
            model.autoencoder.save_weights(os.path.join(path, "autoencoderEpoch%d.pickle" % e))
            model.encoder.save_weights(os.path.join(path, "encoderEpoch%d.pickle" % e))
            model.decoder.save_weights(os.path.join(path, "decoderEpoch%d.pickle" % e))


if __name__ == "__main__":
    main()
