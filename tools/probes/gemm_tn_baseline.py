import os, sys, torch
sys.path.insert(0, "/root/repo")
import midi_vae_amd
from midi_vae_amd import hiplib as hl, ops
dev, bf = "cuda:0", torch.bfloat16
for R in (1048576,):
    H, GH = 256, 1024
    hs = torch.randn((R, H), device=dev).to(bf); da = (torch.randn((R, GH), device=dev) * 0.1).to(bf)
    dU = torch.zeros((H, GH), device=dev)
    for sk in (2, 4, 8, 16, 32, 64):
        fn = lambda: ops.gemm(hs, da, dU, H, GH, R, trans_a=True, accumulate=True, split_k=sk)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("gemm_fast_k dU K=%d split %d: %.3f ms %.1f TFLOP/s" % (R, sk, ms, 2.0 * R * H * GH / ms / 1e9))
