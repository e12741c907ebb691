"""Optimizer kernel alone on a flat buffer the size of BASELINE configs[1]'s parameters (3.6 M f32): 32 bytes of HBM
traffic per parameter (p, g, m, v in; p, m, v and the zeroed g out)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import midi_vae_amd  # noqa: F401
from midi_vae_amd import ops

n = 3_600_000
p, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
t = torch.zeros(1, dtype=torch.int32, device="cuda")
for off in (0, 1):          # 16-byte aligned / unaligned views (scalar path)
    a = [x[off:] for x in (p, g, m, v)]
    for _ in range(5):
        ops.adam_step_dev(*a, 1e-3, t, zero_grad=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    e0.record()
    for _ in range(K):
        ops.adam_step_dev(*a, 1e-3, t, zero_grad=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / K
    print("adam (+ step counter bump), %d parameters, offset %d: %.1f us per step = %.2f TB/s" % (n - off, off, us, (n - off) * 32 / us / 1e6))
