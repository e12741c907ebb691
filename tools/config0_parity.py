#!/usr/bin/env python3
"""BASELINE configs[0] (the reference's CPU-runnable case: seq_len 32 x 4 voices = T 128, z=16, 8 windows, H=256) on the HIP
engine (bf16 resident path and f32 generic path) against the float64 oracle: ELBO and its parts, then three optimizer steps."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec, init_params
from midi_vae_amd.synth import make_windows
from oracle.vae_oracle import OracleVAE, make_cfg

B, T, V, Z = 8, 128, 4, 16
for cell in ("LSTM", "GRU"):
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=2, Le=2, Ld=2)
    params = init_params(spec, 3)
    w = make_windows(B, T, 61, V, 16, 2, Z, seed=1234, epsilon_std=spec.epsilon_std)
    oh = lambda idx, n: np.eye(n)[idx.astype(np.int64)]
    batch = dict(X=oh(w["x_idx"], 61), I=oh(w["i_idx"], 16), Vel=w["vel"][..., None].astype(np.float64),
                 Hist=w["hist"].astype(np.float64), Y=oh(w["x_idx"], 61), C=oh(w["c_idx"], 2))
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    st = orc.new_opt_state(p64)
    want = [orc.train_step(p64, st, batch, w["eps"].astype(np.float64)) for _ in range(3)]
    for dtype in ("bf16", "f32"):
        eng = Engine(spec, max_batch=B, dtype=dtype)
        eng.set_params(params)
        eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
        eng.stage_decoder_inputs(B, hist=w["hist"])
        eng.stage_targets(B, w["x_idx"], w["c_idx"])
        for s in range(3):
            eng.train_step(B)
            m = eng.metrics(B)
            d = abs(m["loss"] - want[s]["loss"])
            print("%s %s step %d: ELBO %.6f (oracle %.6f, |diff| %.2e)  kl %.3e / %.3e  notes %.5f / %.5f" % (
                cell, dtype, s, m["loss"], want[s]["loss"], d, m["kl"], want[s]["kl"], m["notes_loss"], want[s]["notes_loss"]))
            assert d < 1e-3 * (1 + abs(want[s]["loss"])), "ELBO off by more than 1e-3"
print("config 0 parity ok")
