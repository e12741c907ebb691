#!/usr/bin/env python3
"""Time the recurrent kernels alone (T steps, B rows, H=256 bf16) - us per step for every input mode.
   python tools/rnn_microbench.py [--cell LSTM] [--T 512] [--B 256]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa: F401,E402
from midi_vae_amd import hiplib as hl, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="LSTM")
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--rowmajor", action="store_true", help="generic kernels (row-major sequences)")
ap.add_argument("--f32", action="store_true", help="the f32 parity mode: generic kernels, f32 sequences and f32 MFMA")
ap.add_argument("--signal", type=int, default=0, help="publish every N steps in a counter like a pipelined producer (nobody waits)")
ap.add_argument("--concurrent", type=int, default=1, help="run k copies of every launch on k streams (own buffers): contention")
ap.add_argument("--cu-mask", default="", help="streams restricted to a set of CUs (hipExtStreamCreateWithCUMask): 'even' / 'odd' = every "
                "other CU of each shader engine, 'low' = the first half, 'single16' / 'pairs16' = 16 CUs as 16 different CU indices' "
                "first / as 8 neighbouring pairs, or comma-separated hex words")
ap.add_argument("--phased", action="store_true", help="LSTM: phased resident kernels (TILE16) instead of the slot-interleaved ones")
ap.add_argument("--w8", action="store_true", help="GRU: the two-waves-per-SIMD kernels (seq_layout TILE16Q); forward only until the BPTT exists")
a = ap.parse_args()
cell = hl.CELL_CODE[a.cell]
a.rowmajor = a.rowmajor or a.f32
DT = hl.F32 if a.f32 else hl.BF16
LAY = hl.ROWMAJOR if a.rowmajor else (hl.TILE16 if (a.phased or a.cell not in ("LSTM", "GRU")) else hl.TILE16P)
if a.w8:
    LAY = hl.TILE16Q
G, H, T, B = hl.GATES[cell], 256, a.T, a.B
GH = G * H
dev = "cuda:0"
bf = torch.float32 if a.f32 else torch.bfloat16
rng = np.random.default_rng(0)
U = torch.tensor(rng.standard_normal((H, GH)) * 0.03, dtype=torch.float32, device=dev)
up, ut = ops.pack_recurrent(U, cell, DT, 0), ops.pack_recurrent(U, cell, DT, 1)
xp = torch.tensor(rng.standard_normal((T, B, GH)) * 0.5, device=dev).to(bf)
idx = torch.tensor(rng.integers(0, 61, (T, B)), dtype=torch.uint8, device=dev)
table = torch.tensor(rng.standard_normal((61, GH)) * 0.5, device=dev).to(bf)
xs = torch.rand((T, B), device=dev)
w_row, bias = torch.randn(GH, device=dev) * 0.1, torch.randn(GH, device=dev) * 0.1
xp0 = torch.tensor(rng.standard_normal((B, GH)) * 0.5, device=dev).to(bf)
hs = torch.zeros((T + 1, B, H), dtype=bf, device=dev)
cs = torch.zeros((T + 1, B, H), dtype=bf, device=dev) if a.cell == "LSTM" else None
acts = torch.zeros((T, B, GH), dtype=bf, device=dev)
da = torch.zeros((T, B, GH), dtype=bf, device=dev)
rh = torch.zeros((T, B, H), dtype=bf, device=dev)
dext = torch.tensor(rng.standard_normal((T, B, H)) * 0.01, device=dev).to(bf)
hl_ = torch.zeros((B, H), device=dev)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


counters = torch.zeros(1024, dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
sig = dict(chunk_steps=a.signal, signal_done=counters, status=status) if a.signal else {}


def masked_stream(words):
    """a HIP stream whose kernels run only on the CUs of the mask (bit b: XCD b % 8, then shader engine, then CU index - the order
    the driver deals user mask bits out in), wrapped for torch"""
    import ctypes
    path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)
    hip = ctypes.CDLL(path)
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


PRESETS = {"even": [0xFFFFFFFF, 0] * 4, "odd": [0, 0xFFFFFFFF] * 4, "low": [0xFFFFFFFF] * 4 + [0] * 4, "all": [0xFFFFFFFF] * 8,
           "single16": [0xFFFF] + [0] * 7, "pairs16": [0xFF, 0xFF] + [0] * 6, "single32": [0xFFFFFFFF] + [0] * 7,
           "pairs32": [0xFFFF, 0xFFFF] + [0] * 6}
if a.cu_mask:
    words = PRESETS[a.cu_mask] if a.cu_mask in PRESETS else [int(w, 16) for w in a.cu_mask.split(",")]
    streams = [masked_stream(words) for _ in range(a.concurrent)]
else:
    streams = [torch.cuda.Stream() for _ in range(a.concurrent)]
copies = [dict(hs=torch.zeros_like(hs), cs=None if cs is None else torch.zeros_like(cs), acts=torch.zeros_like(acts),
               da=torch.zeros_like(da), rh=torch.zeros_like(rh)) for _ in range(a.concurrent - 1)]


def conc(fn):
    """fn(buffers) on every stream at once (copy 0 uses the shared buffers)"""
    def run():
        if a.concurrent == 1 and not a.cu_mask:
            return fn(None)
        ev = torch.cuda.current_stream().record_event()
        for i, st in enumerate(streams):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                fn(None if i == 0 else copies[i - 1])
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    return run


if a.concurrent > 1 or a.cu_mask:       # what the fork / join harness itself costs per round (a trivial kernel on every stream)
    tiny = torch.zeros(64, device=dev)
    for c in copies:
        c["tiny"] = torch.zeros(64, device=dev)
    HARNESS_MS = timeit(conc(lambda c: (tiny if c is None else c["tiny"]).add_(1.0)))
    print("harness only (one tiny kernel per stream) %7.3f ms per round - included in every line below" % HARNESS_MS)
modes = {"dense": dict(xp=xp), "index": dict(idx=idx, table=table), "scalar": dict(xs=xs, w_row=w_row, bias=bias),
         "const": dict(xp0=xp0)}
flop = 2.0 * B * H * GH * T
for name, kw in modes.items():
    lay = hl.TILE16 if (name == "scalar" and LAY in (hl.TILE16P, hl.TILE16Q)) else LAY
    if name == "index" and a.cell in ("LSTM", "GRU") and lay == hl.TILE16P:      # (random values: any column order times the same)
        kw = dict(kw, table_layout=hl.TABLE_PAIRED)
    if name == "index" and lay == hl.TILE16Q:
        kw = dict(kw, table_layout=hl.TABLE_PAIRED8)
    ms = timeit(conc(lambda c: ops.rnn_fwd(cell, DT, T, B, H, up, hs=c["hs"] if c else hs, cs=c["cs"] if c else cs,
                                           acts=c["acts"] if c else acts, h_last=hl_, seq_layout=lay,
                                           **(sig if lay in (hl.TILE16P, hl.TILE16Q) else {}), **kw)))
    print("fwd %-6s %7.3f ms  %6.2f us/step  %6.1f TFLOP/s" % (name, ms, ms * 1e3 / T, flop / ms / 1e9))
ms = timeit(conc(lambda c: ops.rnn_fwd(cell, DT, T, B, H, up, h_last=hl_, xp0=xp0, seq_layout=LAY)))
print("fwd const (inference, no saves) %7.3f ms  %6.2f us/step" % (ms, ms * 1e3 / T))

def bwd_call(c, ext):
    acts_c, cs_c = (c["acts"], c["cs"]) if c else (acts, cs)
    ops.rnn_bwd(cell, DT, T, B, H, ut, hs, cs_c, acts_c, c["da"] if c else da, dhs_ext=dext if ext else None,
                rh=c["rh"] if c else rh, dh0=hl_, seq_layout=LAY, **(sig if LAY == hl.TILE16P else {}))


for ext in (True, False):
    ms = timeit(conc(lambda c: bwd_call(c, ext)))
    print("bwd ext=%d  %7.3f ms  %6.2f us/step  %6.1f TFLOP/s" % (ext, ms, ms * 1e3 / T, flop / ms / 1e9))
