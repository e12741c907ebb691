#!/usr/bin/env python3
"""BASELINE configs[2] / [3] per-GPU shape (seq_len 256 x 8 voices = T 2048, z=128, 512 windows): three train steps per cell -
buffers beyond 2 GiB, 64 pipeline chunks per stack: finite, falling loss, no pipeline time-out; and the windows/s."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows

B, T, V, Z = 512, 2048, 8, 128
for cell in (sys.argv[1:] or ["LSTM", "GRU"]):
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=2, Le=2, Ld=2)
    eng = Engine(spec, max_batch=B, dtype="bf16", seed=1)
    w = make_windows(B, T, 61, V, 16, 2, Z, seed=7, epsilon_std=spec.epsilon_std)
    eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
    eng.stage_decoder_inputs(B, hist=w["hist"])
    eng.stage_targets(B, w["x_idx"], w["c_idx"])
    losses = []
    for s in range(4):
        if s == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.train_step(B)
        losses.append(eng.metrics(B)["loss"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    eng.check_pipeline()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    print("%s T=%d B=%d z=%d: %.1f ms per step = %.0f windows/s; loss %s; resident %.1f GB" % (
        cell, T, B, Z, dt * 1e3, B / dt, ["%.4f" % x for x in losses], eng.bytes_resident() / 1e9))
    del eng
    torch.cuda.empty_cache()
