#!/usr/bin/env python3
"""End-to-end throughput of the DROP-IN path: windows/s through ``autoencoder.fit`` on reference-format arrays (float64 one-hot
rolls, the lists of vae_definition.prepare_autoencoder_input_and_output_list, reference vae_training.py:802-809) at BASELINE
configs[1] (T=512, z=64, batch 256, LSTM, bf16) - host conversion + upload + train step + history read-back, next to the engine
number of bench.py (inputs resident in HBM).
   python tools/fit_e2e_bench.py [--songs 4] [--windows 1024] [--cell LSTM] [--with-prepass]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, packers as pk
from midi_vae_amd.config import build_settings, create_kwargs
from midi_vae_amd.model import VAE
from midi_vae_amd.synth import make_windows, to_reference_format

ap = argparse.ArgumentParser()
ap.add_argument("--songs", type=int, default=4)
ap.add_argument("--windows", type=int, default=1024, help="windows per song (a fit call = one song)")
ap.add_argument("--cell", default="LSTM")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--with-prepass", action="store_true", help="history pre-pass (encoder.predict, kept on the device) before every fit")
ap.add_argument("--host-history", action="store_true", help="... through host arrays, as the reference does")
ap.add_argument("--lazy", action="store_true", help="read the History objects after the epoch's last song instead of after every fit "
                "(History is filled on first access: the reference-style immediate read waits for the device once per song)")
ap.add_argument("--pace-mask", type=int, default=-1, help="Engine.pace_mask (-1 = as shipped)")
ap.add_argument("--pace-split", type=int, default=-1, help="Engine.pace_mask_split only (-1 = as shipped)")
ap.add_argument("--no-prefetch", action="store_true", help="convert every minibatch on the caller's thread (Autoencoder.prefetch = False)")
ap.add_argument("--threads", type=int, default=-1, help="host packer threads (mvae_host_threads; -1 = default)")
ap.add_argument("--json", action="store_true", help="one JSON line at the end (the last epoch): what bench.py embeds as fit_e2e")
a = ap.parse_args()
import torch
s = build_settings(cell_type=a.cell, input_length=128, output_length=128, latent_dim=64, batch_size=a.batch)
m = VAE().create(compute_dtype="bf16", seed=0, **create_kwargs(s))
if a.no_prefetch:
    m.autoencoder.prefetch = False
n = a.windows
songs = []
for i in range(a.songs):
    w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=10 + i)
    X, Y, C, I, V, D = to_reference_format(w)
    songs.append((X, Y, C, I, V, D, np.zeros((n, s["signature_vector_length"]))))
if a.threads >= 0:
    hl.load().mvae_host_threads(a.threads)
print("host packer threads: %d; one song = %d windows = %.0f MB of float64 one-hot rows (X) + as much again (Y)" % (
    hl.load().mvae_host_threads(-1), n, songs[0][0].nbytes / 1e6))


# host-side wall time inside fit, by part (the device runs asynchronously beside all of it)
from midi_vae_amd import staging as _st, engine as _en
_init = _en.Engine.__init__


def _init_with_knobs(self, *args, **kw):        # (the model builds its engines on first use)
    _init(self, *args, **kw)
    if a.pace_mask >= 0:
        self.pace_mask = self.pace_mask_split = a.pace_mask
    if a.pace_split >= 0:
        self.pace_mask_split = a.pace_split


_en.Engine.__init__ = _init_with_knobs
HOST = {"stage": 0.0, "stage targets": 0.0, "train_step enqueue": 0.0, "read-back": 0.0}


def _timed(cls, name, key):
    fn = getattr(cls, name)

    def wrap(*a_, **k_):
        t = time.perf_counter()
        try:
            return fn(*a_, **k_)
        finally:
            HOST[key] += time.perf_counter() - t
    setattr(cls, name, wrap)


_timed(_st.Stager, "stage", "stage")
_timed(_st.Stager, "finish_targets", "stage targets")
_timed(_en.Engine, "train_step_begin", "train_step enqueue")
_timed(_en.Engine, "train_step_finish", "train_step enqueue")
_timed(_en.Engine, "read_accumulated", "read-back")


def one_song(sg, epoch):
    X, Y, C, I, V, D, S = sg
    t0 = time.perf_counter()
    if a.with_prepass and epoch > 0:
        enc_in = pk.prepare_encoder_input_list(s, X, I, V, D)
        H = m.encoder.predict(enc_in, batch_size=a.batch, device=not a.host_history)
        if a.host_history:
            H = np.concatenate([np.zeros((1, H.shape[1])), H[:-1]])
    else:
        H = np.zeros((n, s["latent_dim"]))
    t1 = time.perf_counter()
    x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, C, I, V, D, S, H, return_sample_weight=True)
    t2 = time.perf_counter()
    h = m.autoencoder.fit(x, y, epochs=1, batch_size=a.batch, shuffle=False, sample_weight=sw, verbose=False)
    loss = h if a.lazy else h.history["loss"][0]
    t3 = time.perf_counter()
    return loss, t1 - t0, t2 - t1, t3 - t2


one_song(songs[0], 0)          # engine construction, first launches
for ep in (1, 2, 3):
    tp = tk = tf = 0.0
    for k_ in HOST:
        HOST[k_] = 0.0
    t0 = time.perf_counter()
    for sg in songs:
        loss, a_, b_, c_ = one_song(sg, ep)
        tp, tk, tf = tp + a_, tk + b_, tf + c_
    if a.lazy:
        loss = loss.history["loss"][0]
    torch.cuda.synchronize()          # (the epoch's wall time ends when the device is done, whatever was read back when)
    dt = time.perf_counter() - t0
    nw = n * len(songs)
    # under --lazy a fit call returns before the device has finished it: its timer is HOST time in fit, not a throughput
    fit_note = ("host time in fit %.2f ms per %d-window step" % (tf / (nw / a.batch) * 1e3, a.batch) if a.lazy else
                "fit alone %.0f windows/s (%.2f ms per %d-window step)" % (nw / tf, tf / (nw / a.batch) * 1e3, a.batch))
    print("epoch %d: %d windows in %.3f s = %.0f windows/s end to end | %s | "
          "pre-pass %.3f s, python packers %.3f s, fit %.3f s | loss %.4f" % (ep, nw, dt, nw / dt, fit_note, tp, tk, tf, loss))
    print("         host time inside fit per %d-window step: %s" % (a.batch, ", ".join(
        "%s %.2f ms" % (k_, v_ / (nw / a.batch) * 1e3) for k_, v_ in HOST.items())))
    last = dict(windows=nw, seconds=dt, windows_per_s=nw / dt, ms_per_optimizer_step=dt / (nw / a.batch) * 1e3,
                fit_only_ms_per_optimizer_step=None if a.lazy else tf / (nw / a.batch) * 1e3, loss=float(loss),
                host_ms_per_step={k_: v_ / (nw / a.batch) * 1e3 for k_, v_ in HOST.items()})
if a.json:
    import json
    print(json.dumps(dict(last, cell=a.cell, batch=a.batch, windows_per_song=n, songs=len(songs), prepass=bool(a.with_prepass),
                          what="autoencoder.fit on the reference's float64 one-hot lists (vae_training.py:802-809), one fit call per "
                               "song: host conversion + upload + train steps + history read-back; third epoch over the songs")))
