#!/usr/bin/env python3
"""Train-step time at BASELINE configs[1] through a ONE-rank RCCL group, with / without the early decoder bucket, under engine
attribute settings:   python tools/dp_knob_bench.py [--overlap 1] [--cell LSTM] knob=value ..."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.dp import make_allreduce
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows
import torch.distributed as dist
ap = argparse.ArgumentParser()
ap.add_argument("--overlap", type=int, default=1)
ap.add_argument("--cell", default="LSTM")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("knobs", nargs="*")
a = ap.parse_args()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
B, T, Z, V = 256, 512, 64, 4
spec = ModelSpec(cell=a.cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=2, Le=2, Ld=2)
w = make_windows(B, T, 61, V, 16, 2, Z, seed=1, epsilon_std=spec.epsilon_std)
eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
for kv in a.knobs:
    k, v = kv.split("=")
    old = getattr(eng, k)
    setattr(eng, k, type(old)(int(v)) if isinstance(old, (bool, int)) else type(old)(v))
eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
eng.stage_decoder_inputs(B, hist=w["hist"])
eng.stage_targets(B, w["x_idx"], w["c_idx"])
hook = make_allreduce(eng, dist, 1, overlap=bool(a.overlap))
for _ in range(30):
    eng.train_step(B, allreduce=hook)
torch.cuda.synchronize()
best = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.train_step(B, allreduce=hook)
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / a.steps * 1e3)
eng.check_pipeline()
print("one-rank RCCL overlap=%d %s %-40s %.3f ms/step (%s)  plans %s" % (
    a.overlap, a.cell, " ".join(a.knobs) or "(defaults)", min(best), " ".join("%.3f" % b for b in best),
    {k: (v if k != "refused" else len(v)) for k, v in eng.plan_stats.items()}))
dist.destroy_process_group()
