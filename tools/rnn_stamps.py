#!/usr/bin/env python3
"""Phase timing of the resident recurrent kernels from s_memtime stamps (library built with -DRES_STAMPS).
   MVAE_LIB=.../libmidivae_hip_stamps.so python tools/rnn_stamps.py --cell GRU --which bwd"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, ops
ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="GRU"); ap.add_argument("--which", default="fwd"); ap.add_argument("--mode", default="dense")
ap.add_argument("--concurrent", type=int, default=1, help="k-1 more copies of the launch on other streams (own output buffers)")
a = ap.parse_args()
cell = hl.CELL_CODE[a.cell]; G, H, T, B = hl.GATES[cell], 256, 512, 256; GH = G * H
dev = "cuda:0"; bf = torch.bfloat16
U = torch.randn((H, GH), device=dev) * 0.03
up, ut = ops.pack_recurrent(U, cell, hl.BF16, 0), ops.pack_recurrent(U, cell, hl.BF16, 1)
xp = (torch.randn((T, B, GH), device=dev) * 0.5).to(bf)
xp0 = (torch.randn((B, GH), device=dev) * 0.5).to(bf)
hs = torch.zeros((T + 1, B, H), dtype=bf, device=dev); cs = torch.zeros((T + 1, B, H), dtype=bf, device=dev) if a.cell == "LSTM" else None
acts = torch.zeros((T, B, GH), dtype=bf, device=dev); da = torch.zeros((T, B, GH), dtype=bf, device=dev)
rh = torch.zeros((T, B, H), dtype=bf, device=dev); dext = (torch.randn((T, B, H), device=dev) * 0.01).to(bf)
hl_ = torch.zeros((B, H), device=dev)
kw = dict(xp=xp) if a.mode == "dense" else dict(xp0=xp0)
streams = [torch.cuda.Stream() for _ in range(a.concurrent)]
bufs = [dict(hs=torch.zeros_like(hs), cs=None if cs is None else torch.zeros_like(cs), acts=torch.zeros_like(acts), da=torch.zeros_like(da),
             rh=torch.zeros_like(rh)) for _ in range(a.concurrent)]
for _ in range(2):
    for st, c in zip(streams, bufs):
        with torch.cuda.stream(st):
            if a.which == "fwd":
                ops.rnn_fwd(cell, hl.BF16, T, B, H, up, hs=c["hs"], cs=c["cs"], acts=c["acts"], h_last=hl_,
                            seq_layout=(2 if a.cell in ("LSTM", "GRU") else 1), **kw)
            else:
                ops.rnn_bwd(cell, hl.BF16, T, B, H, ut, hs, cs, acts, c["da"], dhs_ext=dext, rh=c["rh"], dh0=hl_,
                            seq_layout=(2 if a.cell in ("LSTM", "GRU") else 1))
    torch.cuda.synchronize()
lib = hl.load()
buf = (ctypes.c_ulonglong * 128)()
lib.mvae_debug_stamps.restype = ctypes.c_int
assert lib.mvae_debug_stamps(buf) == 0
st = np.array(buf[:], dtype=np.int64).reshape(8, 16)
print("%s %s %s: cycles relative to stamp 0 of each step (rows = steps 64..71); step length = next row's stamp0 - this" % (a.cell, a.which, a.mode))
for i in range(7):
    rel = [(int(st[i, k] - st[i, 0]) if st[i, k] else -1) for k in range(12)]
    print("step", 64 + i, rel, "len", int(st[i + 1, 0] - st[i, 0]))
