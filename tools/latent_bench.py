#!/usr/bin/env python3
"""Development: the fused latent-chain kernels (csrc/latent.hip) alone, at the bench model's shapes, idle GPU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows

cell = sys.argv[1] if len(sys.argv) > 1 else "LSTM"
Z = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B, T = 256, 512 if Z == 64 else 64
spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=4, ID=16, C=2, Le=2, Ld=2)
eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
w = make_windows(B, T, 61, 4, 16, 2, Z, seed=1, epsilon_std=spec.epsilon_std)
eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
eng.stage_decoder_inputs(B, hist=w["hist"])
eng.stage_targets(B, w["x_idx"], w["c_idx"])
eng.train_step(B)
torch.cuda.synchronize()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(cell, "Z=%d" % Z, end=": "); print("fused fwd   %.1f us" % timed(lambda: eng._latent_chain_forward(B, B, True)))
eng._side = lambda fn: None          # the kernels alone: no parameter-gradient launches
print("fused bwd   %.1f us" % timed(lambda: eng._latent_chain_backward(B, B)))
print("unfused bwd %.1f us (dependent launches on an idle GPU)" % timed(lambda: eng._latent_backward_unfused(B, B)))
