#!/usr/bin/env python3
"""Phase timing of the two-waves-per-SIMD recurrent kernels from s_memtime stamps (library built with -DW8_STAMPS).
   MVAE_W8=1 MVAE_LIB=build/variants/libw8_stamps.so python tools/w8_stamps.py [--mode const|dense] [--save 0|1]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, ops
ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="GRU"); ap.add_argument("--mode", default="const"); ap.add_argument("--save", type=int, default=0)
a = ap.parse_args()
cell = hl.CELL_CODE[a.cell]; G, H, T, B = hl.GATES[cell], 256, 512, 256; GH = G * H
dev = "cuda:0"; bf = torch.bfloat16
U = torch.randn((H, GH), device=dev) * 0.03
up = ops.pack_recurrent(U, cell, hl.BF16, 0)
xp = (torch.randn((T, B, GH), device=dev) * 0.5).to(bf)
xp0 = (torch.randn((B, GH), device=dev) * 0.5).to(bf)
hs = torch.zeros((T + 1, B, H), dtype=bf, device=dev)
acts = torch.zeros((T, B, GH), dtype=bf, device=dev)
hl_ = torch.zeros((B, H), device=dev)
kw = dict(xp=xp) if a.mode == "dense" else dict(xp0=xp0)
if a.save:
    kw.update(hs=hs, acts=acts)
for _ in range(2):
    ops.rnn_fwd(cell, hl.BF16, T, B, H, up, h_last=hl_, seq_layout=hl.TILE16Q, **kw)
    torch.cuda.synchronize()
lib = hl.load()
buf = (ctypes.c_ulonglong * 256)()
lib.mvae_debug_stamps_w8.restype = ctypes.c_int
assert lib.mvae_debug_stamps_w8(buf) == 0
st = np.array(buf[:], dtype=np.int64).reshape(2, 8, 16)
names = {0: "slot0", 2: "pre-2b", 3: "post-2b", 4: "slot15", 5: "slot23", 6: "pre-1", 7: "post-1", 8: "slot31", 9: "slot39", 10: "pre-2a", 11: "post-2a"}
print("%s fwd %s save=%d: cycles relative to the step's slot-0 stamp of wave 0; rows = steps 64..70" % (a.cell, a.mode, a.save))
print("        " + " ".join("%8s" % names[k] for k in sorted(names)) + "      len")
for wv in range(2):
    for i in range(7):
        rel = [int(st[wv, i, k] - st[0, i, 0]) if st[wv, i, k] else -1 for k in sorted(names)]
        print("w%d s%2d " % (wv * 4, 64 + i) + " ".join("%8d" % v for v in rel) + " %8d" % int(st[wv, i + 1, 0] - st[wv, i, 0]))
