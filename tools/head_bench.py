#!/usr/bin/env python3
"""Duration of the notes head (Dense 256->61 + softmax + loss + argmax + d(logits)) over T*B = 131072 rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, ops
dev, bf = "cuda:0", torch.bfloat16
R, H, N, NP = 131072, 256, 61, 64
hs = (torch.randn((R, H), device=dev) * 0.5).to(bf); wt = (torch.randn((NP, H), device=dev) * 0.1).to(bf)
bias = torch.zeros(N, device=dev); tgt = torch.randint(0, N, (R,), device=dev, dtype=torch.uint8)
dl = torch.zeros((R, NP), dtype=bf, device=dev); sc = torch.zeros(2, device=dev); am = torch.zeros(R, dtype=torch.uint8, device=dev)
def run(): ops.head(0, hl.BF16, R, H, N, hs, wt, bias, target_idx=tgt, grad_scale=1.0 / R, argmax=am, dlogits=dl, scalars=sc)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("notes head: %.1f us  (reads %.0f MB, writes %.0f MB -> %.2f TB/s)" % (us, R * H * 2 / 1e6, R * NP * 2 / 1e6, (R * H * 2 + R * NP * 2) / us / 1e6))
wc = (torch.randn((H, NP), device=dev) * 0.1).to(bf); dhs = torch.zeros((R, H), dtype=bf, device=dev)
def run2(): ops.head(0, hl.BF16, R, H, N, hs, wt, bias, target_idx=tgt, grad_scale=1.0 / R, argmax=am, dlogits=dl, scalars=sc, wc=wc, dhs=dhs)
run2(); torch.cuda.synchronize()
e0.record()
for _ in range(20): run2()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("notes head + fused input gradient: %.1f us  (reads %.0f MB, writes %.0f MB -> %.2f TB/s)" % (
    us, R * H * 2 / 1e6, (R * NP * 2 + R * H * 2) / 1e6, (2 * R * H * 2 + R * NP * 2) / us / 1e6))
def run3(): ops.head(0, hl.BF16, R, H, N, hs, wt, bias, target_idx=tgt, grad_scale=1.0 / R, argmax=am, dlogits=dl, scalars=None, wc=wc, dhs=dhs)
run3(); torch.cuda.synchronize()
e0.record()
for _ in range(20): run3()
e1.record(); torch.cuda.synchronize()
print("  the same without the two loss / accuracy atomics per workgroup: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
def run4(): ops.head(0, hl.BF16, R, H, N, hs, wt, bias, target_idx=tgt, grad_scale=1.0 / R, argmax=am, scalars=None)
run4(); torch.cuda.synchronize()
e0.record()
for _ in range(20): run4()
e1.record(); torch.cuda.synchronize()
print("  logits + softmax + argmax only (no gradient, no scalars): %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
x = torch.empty_like(hs)
e0.record()
for _ in range(20): x.copy_(hs)
e1.record(); torch.cuda.synchronize()
print("  (a 67 MB device copy for scale: %.1f us)" % (e0.elapsed_time(e1) / 20 * 1e3))
