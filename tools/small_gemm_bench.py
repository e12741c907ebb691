#!/usr/bin/env python3
"""Durations of the small f32 Dense GEMMs around the latent block (batch 256), one launch each, HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, ops
dev = "cuda:0"
def t(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H, Z, zin, ldS = 256, 256, 64, 128, 2304
r = lambda *s: torch.randn(*s, device=dev)
cases = [
    ("pack   cat(B,768) W(768,256) tanh", lambda: ops.gemm(r(B, 768), r(768, H), torch.empty(B, H, device=dev), B, H, 768, bias=r(H), act=hl.ACT_TANH)),
    ("extra  h(B,256) W(256,256) tanh", lambda: ops.gemm(r(B, H), r(H, H), torch.empty(B, H, device=dev), B, H, H, bias=r(H), act=hl.ACT_TANH)),
    ("zmean  h(B,128) W(128,64)", lambda: ops.gemm(r(B, H), r(128, Z), torch.empty(B, Z, device=dev), B, Z, 128, lda=H, bias=r(Z))),
    ("init   zh(B,128) W(128,2304) tanh", lambda: ops.gemm(r(B, zin), r(zin, ldS), torch.empty(B, ldS, device=dev), B, ldS, zin, bias=r(ldS), act=hl.ACT_TANH)),
    ("xp0    start(B,61) W(61,1024)", lambda: ops.gemm(r(B, 61), r(61, 1024), torch.empty(B, 1024, device=dev), B, 1024, 61, bias=r(1024))),
    ("dzh    dS(B,2304) W^T", lambda: ops.gemm(r(B, ldS), r(zin, ldS), torch.empty(B, zin, device=dev), B, zin, ldS, trans_b=True)),
    ("dinitW zh^T dS (acc)", lambda: ops.gemm(r(B, zin), r(B, ldS), torch.zeros(zin, ldS, device=dev), zin, ldS, B, trans_a=True, accumulate=True)),
    ("dt     dmu(B,64) W^T(64->128)", lambda: ops.gemm(r(B, Z), r(128, Z), torch.empty(B, H, device=dev), B, 128, Z, trans_b=True, ldc=H)),
    ("dcat   dt(B,256) W^T(->768)", lambda: ops.gemm(r(B, H), r(768, H), torch.empty(B, 768, device=dev), B, 768, H, trans_b=True)),
    ("dpackW cat^T dt (acc)", lambda: ops.gemm(r(B, 768), r(B, H), torch.zeros(768, H, device=dev), 768, H, B, trans_a=True, accumulate=True)),
    ("colsum dS (B,2304) f32", lambda: ops.colsum(r(B, ldS), B, ldS, torch.zeros(ldS, device=dev))),
    ("tanh_bwd (B,2304)", lambda: ops.tanh_bwd(r(B, ldS), r(B, ldS), torch.empty(B, ldS, device=dev))),
]
# inputs are created inside the lambdas (torch.randn kernels): subtract their cost measured alone
base = {}
for name, fn in cases:
    print("%-36s %7.1f us (incl. input generation)" % (name, t(fn)))
print("randn(B,2304)+randn(128,2304) alone   %7.1f us" % t(lambda: (r(B, ldS), r(zin, ldS))))
