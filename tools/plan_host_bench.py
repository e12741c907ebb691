#!/usr/bin/env python3
"""Host enqueue time of a train step against its device time, Python enqueue vs step-plan replay (include/midivae_hip.h 'STEP
PLANS'), at BASELINE configs[1] and at the reference's shipped configuration (settings.py:108-112,140,155: GRU, T=64, Z=256,
batch 256).      python tools/plan_host_bench.py [--shape bench|reference] [--cell LSTM|GRU] [--steps 50]
Every mode runs in a process of its own (an engine that is not the first of its process runs slower).  Two host figures:
the ENQUEUE alone (pauses off, an idle device in front of every step: what the CPU needs) and the time spent inside train_step()
back to back as shipped - which includes the paced host's waits for the device and says nothing about who is the bottleneck."""
import argparse, os, subprocess, sys, time
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="bench", choices=["bench", "reference"])
ap.add_argument("--cell", default=None)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--plans", type=int, default=-1, help="(internal) 0 / 1: one mode, in this process")
a = ap.parse_args()
if a.plans < 0:
    for plans in (0, 1):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--shape", a.shape, "--steps", str(a.steps), "--plans", str(plans)] +
                       (["--cell", a.cell] if a.cell else []), check=True)
    sys.exit(0)
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows
if a.shape == "bench":
    cell, T, Z, B, V = a.cell or "LSTM", 512, 64, 256, 4
else:
    cell, T, Z, B, V = a.cell or "GRU", 64, 256, 256, 4
spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=2, Le=2, Ld=2)
w = make_windows(B, T, 61, V, 16, 2, Z, seed=1, epsilon_std=spec.epsilon_std)
plans = bool(a.plans)
if not plans:
    print("%s shape: %s T=%d Z=%d B=%d" % (a.shape, cell, T, Z, B))
eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
eng.use_plans = plans
eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
eng.stage_decoder_inputs(B, hist=w["hist"])
eng.stage_targets(B, w["x_idx"], w["c_idx"])
def warm(n):
    for _ in range(n):
        eng.train_step(B)
    torch.cuda.synchronize()
warm(20)
# (1) the enqueue alone: pauses off, the device idle in front of every step (sync before, not inside)
shipped = (eng.pace_mask, eng.steps_in_flight)
eng.pace_mask, eng.steps_in_flight = 0, 0
warm(6)
host = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_step(B)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
eng.pace_mask, eng.steps_in_flight = shipped
warm(6)
# (2) as shipped, steps back to back: wall per step and the time spent inside train_step()
t0 = time.perf_counter()
for _ in range(a.steps):
    eng.train_step(B)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
eng.check_pipeline()
wall, inside, enq = (t2 - t0) / a.steps * 1e3, (t1 - t0) / a.steps * 1e3, float(np.median(host))
print("  %-14s enqueue alone %.3f ms/step (median of 10, pauses off)   back to back as shipped: %.3f ms/step wall [%d windows/s], "
      "%.3f ms inside train_step() (the paced host waits there) -> %s" % (
          "plan replay" if plans else "python enqueue", enq, wall, B * a.steps / (t2 - t0), inside,
          "device-bound: the enqueue is %.0f %% of a step" % (100 * enq / wall) if enq < 0.8 * wall else "HOST-bound: the enqueue alone is %.0f %% of a step" % (100 * enq / wall)))
if plans:
    print("  plan stats:", {k: (v if k != "refused" else len(v)) for k, v in eng.plan_stats.items()})
