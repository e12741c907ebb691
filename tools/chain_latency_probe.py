#!/usr/bin/env python3
"""Latency of a chain of dependent tiny kernels: eager launches vs one hipGraph replay (same kernels, same stream)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, ops
dev = "cuda:0"
B, H = 256, 256
x = torch.randn(B, H, device=dev); W = [torch.randn(H, H, device=dev) * 0.05 for _ in range(9)]
bias = torch.zeros(H, device=dev); bufs = [torch.empty(B, H, device=dev) for _ in range(10)]
def chain():
    src = x
    for i in range(9):
        ops.gemm(src, W[i], bufs[i], B, H, H, bias=bias, act=hl.ACT_TANH)
        src = bufs[i]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("eager chain of 9 dependent 256x256x256 f32 GEMMs: %.1f us" % timeit(chain))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain()
torch.cuda.synchronize()
print("graph replay of the same chain:                  %.1f us" % timeit(g.replay))
