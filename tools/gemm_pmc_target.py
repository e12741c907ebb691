#!/usr/bin/env python3
"""Development: ten launches of the weight-gradient GEMM dU = hs^T da (256 x 1024, K = 131072, split-K 16) for rocprofv3 --pmc."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import ops
dev, bf = "cuda:0", torch.bfloat16
R, H, GH = 512 * 256, 256, 1024
hs = torch.randn((R, H), device=dev).to(bf); da = (torch.randn((R, GH), device=dev) * 0.1).to(bf)
dU = torch.zeros((H, GH), device=dev)
for _ in range(10):
    ops.gemm(hs, da, dU, H, GH, R, trans_a=True, accumulate=True, split_k=16)
torch.cuda.synchronize()
