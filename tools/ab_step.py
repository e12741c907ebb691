#!/usr/bin/env python3
"""Development: A/B of engine switches on the bench workload, same process, alternating.
   python tools/ab_step.py attr=value[,attr=value...] [more variants ...]   (first variant '-' = defaults)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows

cell = os.environ.get("CELL", "LSTM")
B, T = 256, 512
spec = ModelSpec(cell=cell, H=256, Z=64, Din=61, Dout=61, T=T, V=4, ID=16, C=2, Le=2, Ld=2)
w = make_windows(B, T, 61, 4, 16, 2, 64, seed=1, epsilon_std=spec.epsilon_std)
engines = []
for var in sys.argv[1:] or ["-"]:
    eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
    if var != "-":
        for kv in var.split(","):
            k, v = kv.split("=")
            setattr(eng, k, type(getattr(eng, k))(eval(v)))
    eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
    eng.stage_decoder_inputs(B, hist=w["hist"])
    eng.stage_targets(B, w["x_idx"], w["c_idx"])
    for _ in range(3):
        eng.train_step(B)
    engines.append((var, eng))
torch.cuda.synchronize()
for rep in range(3):
    for var, eng in engines:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.train_step(B)
        torch.cuda.synchronize()
        print("%-50s %.3f ms/step" % (var, (time.perf_counter() - t0) * 100))
