#!/usr/bin/env python3
"""Host enqueue time of a train step against its device time, Python enqueue vs step-plan replay (include/midivae_hip.h 'STEP
PLANS'), at BASELINE configs[1] and at the reference's shipped configuration (settings.py:108-112,140,155: GRU, T=64, Z=256,
batch 256).      python tools/plan_host_bench.py [--shape bench|reference] [--cell LSTM|GRU] [--steps 50]"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="bench", choices=["bench", "reference"])
ap.add_argument("--cell", default=None)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
if a.shape == "bench":
    cell, T, Z, B, V = a.cell or "LSTM", 512, 64, 256, 4
else:
    cell, T, Z, B, V = a.cell or "GRU", 64, 256, 256, 4
spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=2, Le=2, Ld=2)
w = make_windows(B, T, 61, V, 16, 2, Z, seed=1, epsilon_std=spec.epsilon_std)
print("%s shape: %s T=%d Z=%d B=%d" % (a.shape, cell, T, Z, B))
for plans in (False, True):
    eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
    eng.use_plans = plans
    eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
    eng.stage_decoder_inputs(B, hist=w["hist"])
    eng.stage_targets(B, w["x_idx"], w["c_idx"])
    for _ in range(20):
        eng.train_step(B)
    torch.cuda.synchronize()
    # (1) host alone: enqueue time of one step with an idle device in front of it (sync before every step)
    host = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train_step(B)
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    # (2) steps back to back: wall per step, and how long the host spent enqueueing them
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.train_step(B)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    eng.check_pipeline()
    print("  %-14s host enqueue %.3f ms/step (median of 10, device idle)   back to back: %.3f ms/step wall, host busy %.3f ms/step "
          "-> %s   [%s]" % ("plan replay" if plans else "python enqueue", float(np.median(host)), (t2 - t0) / a.steps * 1e3,
                            (t1 - t0) / a.steps * 1e3, "host-bound" if (t1 - t0) > 0.9 * (t2 - t0) else "device-bound",
                            "%d windows/s" % (B * a.steps / (t2 - t0))))
    if plans:
        print("  plan stats:", eng.plan_stats)
    del eng
    torch.cuda.empty_cache()
