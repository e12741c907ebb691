"""host time per Engine.decode call with / without device-side joins (development: why decode slows down with MVAE_DEVICE_JOIN=1)"""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
spec = ModelSpec(cell="LSTM", H=256, Z=64, Din=61, Dout=61, T=512, V=4, ID=16, C=2, Le=2, Ld=2)
eng = Engine(spec, max_batch=256, dtype="bf16", device="cuda:0", seed=1234, training=False)
rng = np.random.default_rng(1); z = rng.standard_normal((256, 64)).astype(np.float32)
eng.stage_decoder_inputs(256, hist=z, z=z)
for _ in range(3):
    eng.decode(256, want_probs=False)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(40):
    a = time.perf_counter(); eng.decode(256, want_probs=False); host.append(time.perf_counter() - a)
torch.cuda.synchronize()
print("device_join", eng.device_join, "wall per batch %.2f ms; host per call median %.2f ms max %.2f ms" % (
    (time.perf_counter() - t0) / 40 * 1e3, np.median(host) * 1e3, max(host) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    eng.decode(256, want_probs=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
