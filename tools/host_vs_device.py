#!/usr/bin/env python3
"""Is the host ahead of the device at every section boundary of a training step?  Steps run back to back (as in bench.py); for
one step in the middle: when each boundary marker was ENQUEUED by the host and when the device REACHED it, on one clock.
   python tools/host_vs_device.py [--cell LSTM] [--steps 12]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows
ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="LSTM"); ap.add_argument("--steps", type=int, default=12)
a = ap.parse_args()
T, B = 512, 256
spec = ModelSpec(cell=a.cell, H=256, Z=64, Din=61, Dout=61, T=T, V=4, ID=16, C=2, Le=2, Ld=2)
eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
w = make_windows(B, T, 61, 4, 16, 2, 64, seed=1, epsilon_std=spec.epsilon_std)
eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
eng.stage_decoder_inputs(B, hist=w["hist"])
eng.stage_targets(B, w["x_idx"], w["c_idx"])
for _ in range(30):
    eng.train_step(B)
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True)
hb = time.perf_counter()
base.record()
eng.marks = []
for _ in range(a.steps):
    eng.train_step(B)
    eng._mark("optimizer enqueued")
torch.cuda.synchronize()
marks, eng.marks = eng.marks, None
rows = [(n, (h - hb) * 1e3, base.elapsed_time(e)) for n, h, e in marks]
starts = [i for i, r in enumerate(rows) if r[0] == "step start"]
k = starts[len(starts) // 2]
k2 = starts[len(starts) // 2 + 1]
h0, d0 = rows[k][1], rows[k][2]
print("step %d of %d (ms; host = marker enqueued, device = marker reached; lead = device - host on the common clock)" % (len(starts) // 2, a.steps))
for n, h, d in rows[k:k2 + 1]:
    print("  %-34s host %8.3f   device %8.3f   lead %7.3f" % (n, h - h0, d - d0, d - h))
print("host enqueue of the step %.3f ms, device step %.3f ms" % (rows[k2][1] - h0, rows[k2][2] - d0))
