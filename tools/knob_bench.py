#!/usr/bin/env python3
"""Train-step time of one shape under engine attribute settings (schedule knobs that are plain attributes, not environment
switches): pure plan replay, back to back.
   python tools/knob_bench.py --shape reference [--cell GRU] pipe_chunk=8 defer_grads_rows=0 ..."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="reference", choices=["bench", "reference", "config2"])
ap.add_argument("--cell", default=None)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--inflight", type=int, default=0, help="at most this many steps enqueued ahead of the device (0 = no limit)")
ap.add_argument("--bracket", type=int, default=0, help="every Nth step carries HIP-event brackets around the BPTT launches (bench.py's measurement); "
                "reports the bracketed and the plain steps apart (per-step synchronisation)")
ap.add_argument("--no-meta", action="store_true", help="no velocity / instrument branches (encoder rolls and decoder heads)")
ap.add_argument("--layers", type=int, default=2, help="cells per stack (encoder and decoder)")
ap.add_argument("--skip-streams", type=int, default=0, help="take this many streams out of torch's pool before the engine is built")
ap.add_argument("knobs", nargs="*")
a = ap.parse_args()
C = 2
if a.shape == "bench":
    cell, T, Z, V = a.cell or "LSTM", 512, 64, 4
elif a.shape == "config2":            # BASELINE configs[2] / [3] per-GPU shape: seq 256 x 8 voices, z=128, 4 styles, 512 windows
    cell, T, Z, V, C = a.cell or "LSTM", 2048, 128, 8, 4
    if a.batch == 256:
        a.batch = 512
else:
    cell, T, Z, V = a.cell or "GRU", 64, 256, 4
B = a.batch
spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=C, Le=a.layers, Ld=a.layers,
                 **(dict(meta_velocity=False, meta_instrument=False) if a.no_meta else {}))
w = make_windows(B, T, 61, V, 16, C, Z, seed=1, epsilon_std=spec.epsilon_std)
_skipped = [torch.cuda.Stream(device="cuda:0") for _ in range(a.skip_streams)]
eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
for kv in a.knobs:
    k, v = kv.split("=")
    old = getattr(eng, k)
    setattr(eng, k, type(old)(int(v)) if isinstance(old, (bool, int)) else type(old)(v))
eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
eng.stage_decoder_inputs(B, hist=w["hist"])
eng.stage_targets(B, w["x_idx"], w["c_idx"])
for _ in range(30):
    eng.train_step(B)
torch.cuda.synchronize()
best, host, dev = [], [], []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    ring = []
    for _ in range(a.steps):
        if a.inflight:
            if len(ring) >= a.inflight:
                ring.pop(0).synchronize()
            ev = torch.cuda.Event()
        eng.train_step(B)
        if a.inflight:
            ev.record()
            ring.append(ev)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / a.steps * 1e3)
    host.append((t1 - t0) / a.steps * 1e3)
    dev.append(e0.elapsed_time(e1) / a.steps)
if a.bracket:
    kinds = {("rnn_bwd", "dec.notes.1"), ("rnn_bwd", "dec.notes.0"), ("rnn_bwd_multi", "dec")}
    eng.prof_kinds = kinds
    per = {True: [], False: []}
    hostt = {True: [], False: []}
    prof = {}
    for i in range(a.steps + 16):
        br = i % a.bracket == 0
        eng.prof = prof if br else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train_step(B)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if i >= 16:
            per[br].append((time.perf_counter() - t0) * 1e3)
            hostt[br].append((t1 - t0) * 1e3)
    eng.prof = None
    med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
    print("bracket every %d: bracketed steps median %.3f ms (host %.3f, max %.3f, n %d) | plain steps median %.3f ms (host %.3f, max %.3f, n %d)" % (
        a.bracket, med(per[True]), med(hostt[True]), max(per[True]), len(per[True]), med(per[False]), med(hostt[False]),
        max(per[False] or [0]), len(per[False])))
eng.check_pipeline()
print("%-9s %s T=%d B=%d %-40s %.3f ms/step (3 x %d steps: %s; host enqueue %.3f, events %.3f)  [%d windows/s]  plans %s" % (
    a.shape, cell, T, B, (" ".join(a.knobs) or "(defaults)") + (" no-meta" if a.no_meta else "") + (" layers=%d" % a.layers if a.layers != 2 else "") + (" skip=%d" % a.skip_streams if a.skip_streams else "") + (" inflight=%d" % a.inflight if a.inflight else ""), min(best), a.steps, " ".join("%.3f" % b for b in best), min(host), min(dev),
    B / min(best) * 1e3,
    {k: (v if k != "refused" else len(v)) for k, v in eng.plan_stats.items()}))
