#!/usr/bin/env python3
"""The drop-in training script at the reference's DEFAULT settings, end to end: ``vae_training.run_epoch``
(reference vae_training.py:775-815) over synthetic songs of 20-200 windows - GRU, T=64, Z=256, batch 256 (settings.py:108-112,140,
155) - epoch 0 without the history pre-pass, epochs 1.. with it (encoder.predict on the device, fused into fit).  Reports windows/s
per epoch, the host time inside fit by part and the device-busy share (HIP events around every epoch's enqueue).
   python tools/training_script_bench.py [--songs 64] [--epochs 4]"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import engine as _en, staging as _st
from midi_vae_amd.config import build_settings, create_kwargs
from midi_vae_amd.synth import make_windows, to_reference_format
import vae_training
from vae_definition import VAE

ap = argparse.ArgumentParser()
ap.add_argument("--songs", type=int, default=64)
ap.add_argument("--epochs", type=int, default=4)
ap.add_argument("--min-windows", type=int, default=20)
ap.add_argument("--max-windows", type=int, default=200)
ap.add_argument("--pace-mask", type=int, default=-1, help="Engine.pace_mask (-1 = as shipped)")
ap.add_argument("--pace-split", type=int, default=-1, help="Engine.pace_mask_split only (-1 = as shipped)")
ap.add_argument("--in-flight", type=int, default=-1, help="Engine.steps_in_flight (-1 = as shipped)")
a = ap.parse_args()
s = build_settings()
m = VAE().create(compute_dtype="bf16", seed=0, **create_kwargs(s))
_init = _en.Engine.__init__


def _init_with_knobs(self, *args, **kw):        # (the model builds its engines on first use)
    _init(self, *args, **kw)
    if a.pace_mask >= 0:
        self.pace_mask = self.pace_mask_split = a.pace_mask
    if a.pace_split >= 0:
        self.pace_mask_split = a.pace_split
    if a.in_flight >= 0:
        self.steps_in_flight = a.in_flight


_en.Engine.__init__ = _init_with_knobs
rng = np.random.default_rng(7)
songs = []
for i in range(a.songs):
    n = int(rng.integers(a.min_windows, a.max_windows + 1))
    w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], s["meta_instrument_dim"], s["num_classes"], s["latent_dim"],
                     seed=7000 + i)
    X, Y, _, I, V, D = to_reference_format(w, s["output_dim"], s["meta_instrument_dim"])
    songs.append(dict(X=X, Y=Y, C=i % s["num_classes"], I=I, V=V, D=D, S=np.zeros((n, s["signature_vector_length"]))))
nw = sum(sg["X"].shape[0] for sg in songs)
steps = sum(-(-sg["X"].shape[0] // s["batch_size"]) for sg in songs)
print("default settings: %s T=%d Z=%d batch %d; %d songs of %d-%d windows = %d windows, %d optimizer steps per epoch (mean %.0f windows per step)"
      % (s["cell_type"], s["output_length"], s["latent_dim"], s["batch_size"], a.songs, a.min_windows, a.max_windows, nw, steps, nw / steps))

HOST = {"stage": 0.0, "stage targets": 0.0, "train_step enqueue (incl. pacing waits)": 0.0, "read-back": 0.0}


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
_timed(_en.Engine, "train_step_begin", "train_step enqueue (incl. pacing waits)")
_timed(_en.Engine, "train_step_finish", "train_step enqueue (incl. pacing waits)")
_timed(_en.Engine, "read_accumulated", "read-back")
for ep in range(a.epochs):
    for k in HOST:
        HOST[k] = 0.0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    out = vae_training.run_epoch(m, songs, s, ep, train=True)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("epoch %d (%s): %d windows in %.3f s = %.0f windows/s, %.2f ms per optimizer step end to end (host returned after %.3f s); loss %.4f"
          % (ep, "no pre-pass" if ep == 0 else "history pre-pass on the device", nw, dt, nw / dt, dt / steps * 1e3, t1 - t0, out["loss"]))
    print("         host time inside fit per optimizer step: %s" % ", ".join("%s %.2f ms" % (k, v / steps * 1e3) for k, v in HOST.items()))
eng = m._shared.engine
print("pace_mask %d, steps_in_flight %d" % (eng.pace_mask, eng.steps_in_flight))
eng.check_pipeline()
print("plans:", {k: (v if k != "refused" else len(v)) for k, v in eng.plan_stats.items()})
