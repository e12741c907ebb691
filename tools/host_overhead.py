#!/usr/bin/env python3
"""Host-side enqueue time of one training step vs its device time (is the step launch-bound?).
   python tools/host_overhead.py [--cell LSTM] [--chunks 4]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows
ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="LSTM"); ap.add_argument("--chunks", type=int, default=0)
ap.add_argument("--no-side-grads", action="store_true");
ap.add_argument("--pipe-chunk", type=int, default=-1, help="0 = pipeline off");
ap.add_argument("--grad-blocks", type=int, default=0); ap.add_argument("--no-chunk-grads", action="store_true"); ap.add_argument("--one-grad-stream", action="store_true")
a = ap.parse_args()
T, B = 512, 256
spec = ModelSpec(cell=a.cell, H=256, Z=64, Din=61, Dout=61, T=T, V=4, ID=16, C=2, Le=2, Ld=2)
eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
if a.chunks:
    eng.time_chunks = a.chunks
eng.grad_gemm_blocks = a.grad_blocks
if a.pipe_chunk == 0:
    eng.pipeline = False
elif a.pipe_chunk > 0:
    eng.pipe_chunk = a.pipe_chunk
eng.side_grads = not a.no_side_grads
if a.no_chunk_grads:
    eng.grad_per_chunk = False
if a.one_grad_stream:
    eng.s_grad2 = eng.s_grad
w = make_windows(B, T, 61, 4, 16, 2, 64, seed=1, epsilon_std=spec.epsilon_std)
eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
eng.stage_decoder_inputs(B, hist=w["hist"])
eng.stage_targets(B, w["x_idx"], w["c_idx"])
for _ in range(3):
    eng.train_step(B)
torch.cuda.synchronize()
host, dev = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_step(B)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); dev.append((t2 - t0) * 1e3)
print("host enqueue ms/step:", ["%.2f" % h for h in host])
print("enqueue + drain ms/step:", ["%.2f" % d for d in dev])

torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eng.train_step(B)
torch.cuda.synchronize()
print("10 steps back to back: %.2f ms/step" % ((time.perf_counter() - t0) * 100))
eng.prof, eng.prof_kinds = {}, {"rnn_bwd"}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eng.train_step(B)
torch.cuda.synchronize()
print("10 steps back to back, BPTT launches bracketed with events: %.2f ms/step" % ((time.perf_counter() - t0) * 100))
eng.prof = None
# section times of one step (events on the main stream at section boundaries)
eng.marks = []
eng.train_step(B)
e_end = torch.cuda.Event(enable_timing=True); e_end.record()
torch.cuda.synchronize()
prev = eng.marks[0][1]
for name, e in eng.marks[1:] + [("optimizer", e_end)]:
    print("%-36s %7.3f ms" % (name, prev.elapsed_time(e)) if not name.startswith("  ") else "%-36s   (at +%.3f ms)" % (name, eng.marks[0][1].elapsed_time(e)))
    if not name.startswith("  "):
        prev = e
eng.marks = None

# the same with 4 steps queued back to back (no host synchronisation in between): sections of the last one
eng.marks = []
for _ in range(4):
    eng.train_step(B)
e_end = torch.cuda.Event(enable_timing=True); e_end.record()
torch.cuda.synchronize()
n = len(eng.marks) // 4
last = eng.marks[-n:]
prev = last[0][1]
print("back to back, last of 4 steps:")
for name, e in last[1:] + [("optimizer", e_end)]:
    if not name.startswith("  "):
        print("%-36s %7.3f ms" % (name, prev.elapsed_time(e)))
        prev = e
    else:
        print("%-36s   (at +%.3f ms)" % (name, last[0][1].elapsed_time(e)))
eng.marks = None
