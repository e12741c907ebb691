#!/usr/bin/env python3
"""Decoder inference through the PRODUCT path: ``decoder.predict_note_indices`` on the reference's list layout (reference
vae_evaluation.py:2471-2483: latent swap, decoder.predict, argmax) - host lists in, (n, T) note indices out - beside the engine
figure of tools/decode_bench.py / bench.py --config 4 (inputs resident in HBM).  The per-GPU share of BASELINE configs[4]:
T = 4096, z = 128, 1024 windows.
   python tools/decode_product_bench.py [--windows 1024] [--cell LSTM]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import packers as pk
from midi_vae_amd.config import build_settings, create_kwargs
from midi_vae_amd.model import VAE
ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=1024)
ap.add_argument("--cell", default="LSTM")
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
import torch
s = build_settings(cell_type=a.cell, input_length=512, output_length=512, max_voices=8, latent_dim=128, batch_size=256)
m = VAE().create(compute_dtype="bf16", seed=0, **create_kwargs(s))
n = a.windows
rng = np.random.default_rng(0)
z = rng.standard_normal((n, s["latent_dim"]))
z[:, [0, 1]] = z[:, [1, 0]]
S = np.zeros((n, s["signature_vector_length"]))
dec_in = pk.prepare_decoder_input(s, z, 0, S, None)
idx = m.decoder.predict_note_indices(dec_in, batch_size=256)          # engine construction, first launches
ts = []
for _ in range(a.reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = m.decoder.predict_note_indices(dec_in, batch_size=256)
    ts.append(time.perf_counter() - t0)
dt = min(ts)
print("decoder.predict_note_indices (%s, T=%d, z=%d, %d windows, caller batch_size 256 -> forward-only engine of %d windows): %.2f ms = "
      "%.0f windows/s end to end (host lists in, (n, T) uint8 indices out; best of %d: %s ms); indices %s" % (
          a.cell, s["output_length"], s["latent_dim"], n, m._shared.infer.maxB, dt * 1e3, n / dt, a.reps,
          ", ".join("%.1f" % (t * 1e3) for t in ts), tuple(idx.shape)))
