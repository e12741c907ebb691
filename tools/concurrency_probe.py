#!/usr/bin/env python3
"""Does a resident LSTM recurrence slow down when other recurrences run beside it on other CUs?
   N copies of the same forward launch (own buffers) on N streams; time from first launch to all done."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, ops
cell, G, H, T, B = hl.LSTM, 4, 256, 512, 256
GH = G * H
dev, bf = "cuda:0", torch.bfloat16
def mk():
    U = torch.randn((H, GH), device=dev) * 0.03
    return dict(up=ops.pack_recurrent(U, cell, hl.BF16, 0), ut=ops.pack_recurrent(U, cell, hl.BF16, 1),
                xp=(torch.randn((T, B, GH), device=dev) * 0.5).to(bf), hs=torch.zeros((T + 1, B, H), dtype=bf, device=dev),
                cs=torch.zeros((T + 1, B, H), dtype=bf, device=dev), acts=torch.zeros((T, B, GH), dtype=bf, device=dev),
                da=torch.zeros((T, B, GH), dtype=bf, device=dev), dext=(torch.randn((T, B, H), device=dev) * 0.01).to(bf),
                hl=torch.zeros((B, H), device=dev))
sets = [mk() for _ in range(6)]
streams = [torch.cuda.Stream() for _ in range(6)]
def fwd(s): ops.rnn_fwd(cell, hl.BF16, T, B, H, s["up"], xp=s["xp"], hs=s["hs"], cs=s["cs"], acts=s["acts"], h_last=s["hl"], seq_layout=hl.TILE16P)
def bwd(s): ops.rnn_bwd(cell, hl.BF16, T, B, H, s["ut"], s["hs"], s["cs"], s["acts"], s["da"], dhs_ext=s["dext"], dh0=s["hl"], seq_layout=hl.TILE16P)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for n in (() if os.environ.get("NEIGHBOURS_ONLY") else (1, 2, 3, 4, 6)):
        for _ in range(2):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                streams[i].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(streams[i]):
                    fn(sets[i])
            for i in range(n):
                torch.cuda.current_stream().wait_stream(streams[i])
            e1.record()
            torch.cuda.synchronize()
        print("%s x%d concurrent: %.3f ms  (%.2f us per time step each)" % (name, n, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / T))

# One recurrence beside a memory streamer (device-to-device copies of 1 GiB on another stream: ~HBM speed, thrashes every
# L2) and beside a compute-only neighbour (a GEMM whose operands fit in L2): which of the two slows it down?
big_a = torch.empty(1 << 29, dtype=torch.int16, device=dev); big_b = torch.empty_like(big_a)
sa = (torch.randn((4096, 1024), device=dev)).to(bf); sb = (torch.randn((1024, 1024), device=dev)).to(bf); sc = torch.zeros((4096, 1024), dtype=bf, device=dev)
def copies():
    for _ in range(12): big_b.copy_(big_a)
def small_gemms():
    for _ in range(200): ops.gemm(sa, sb, sc, 4096, 1024, 1024, trans_b=True)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for bg_name, bg in (("alone", None), ("+ 1 GiB copies (HBM / L2 streamer)", copies), ("+ L2-resident GEMMs (compute neighbour)", small_gemms)):
        for _ in range(2):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if bg is not None:
                streams[1].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(streams[1]):
                    bg()
            with torch.cuda.stream(streams[0]):
                e0.record(); fn(sets[0]); e1.record()
            torch.cuda.synchronize()
        print("%s %-45s %.3f ms  (%.2f us per time step)" % (name, bg_name, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / T))
