#!/usr/bin/env python3
"""GEMM shapes of the train step (BASELINE configs[1]) - TFLOP/s per variant."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl, ops
dev, bf = "cuda:0", torch.bfloat16
R, H, GH = 512 * 256, 256, 1024
hs = torch.randn((R, H), device=dev).to(bf); da = (torch.randn((R, GH), device=dev) * 0.1).to(bf)
wt = torch.randn((GH, H), device=dev).to(bf); wc = torch.randn((H, GH), device=dev).to(bf)
xp = torch.zeros((R, GH), dtype=bf, device=dev); dx = torch.zeros((R, H), dtype=bf, device=dev)
dU = torch.zeros((H, GH), device=dev); idx = torch.randint(0, 61, (R,), dtype=torch.uint8, device=dev)
dW = torch.zeros((61, GH), device=dev); bias = torch.zeros(GH, device=dev)
def t(fn, flop, name, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-44s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flop / ms / 1e9))
f = 2.0 * R * H * GH
t(lambda: ops.gemm(hs, wt, xp, R, GH, H, trans_b=True, bias=bias, c_layout=hl.TILE16), f, "xp = hs W^T (NT, tile16 out)")
t(lambda: ops.gemm(da, wc, dx, R, H, GH, trans_b=True, c_layout=hl.TILE16), f, "dx = da W (NT, K=1024)")
t(lambda: ops.gemm(hs, da, dU, H, GH, R, trans_a=True, accumulate=True, split_k=64), f, "dU = hs^T da (TN split-K 64)")
t(lambda: ops.gemm(hs, da, dU, H, GH, R, trans_a=True, accumulate=True, split_k=16), f, "dU = hs^T da (TN split-K 16)")
t(lambda: ops.gemm(idx, da, dW, 61, GH, R, trans_a=True, a_kind=hl.ONEHOT, accumulate=True, split_k=64), 2.0 * R * 61 * GH, "dWtab = onehot^T da (split-K 64)")
for sk in (8, 16, 32):
    t(lambda: ops.gemm(hs, da, dU, H, GH, R, trans_a=True, accumulate=True, split_k=sk), f, "dU = hs^T da (TN split-K %d)" % sk)
# two at a time on two streams, as in the step
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
dU2 = torch.zeros((H, GH), device=dev)
def pair():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): ops.gemm(hs, da, dU, H, GH, R, trans_a=True, accumulate=True, split_k=16)
    with torch.cuda.stream(s2): ops.gemm(hs, da, dU2, H, GH, R, trans_a=True, accumulate=True, split_k=16)
    cur.wait_stream(s1); cur.wait_stream(s2)
t(pair, 2 * f, "two dU GEMMs on two streams (split-K 16)")
hsT = hs.t().contiguous()
t(lambda: ops.gemm(hsT, da, dU, H, GH, R, accumulate=True, split_k=16), f, "dU = (hs^T stored k-contiguous) da (NN split-K 16)")
daT = da.t().contiguous()
t(lambda: ops.gemm(hsT, daT, dU, H, GH, R, trans_b=True, accumulate=True, split_k=16), f, "dU, both operands k-contiguous (NT split-K 16)")
cs = torch.zeros((GH,), device=dev)
t(lambda: ops.gemm(hs, da, dU, H, GH, R, trans_a=True, accumulate=True, split_k=16, colsum_b=cs), f, "dU + column sums of da (TN split-K 16)")
# ---- the persistent / waiting variants of the step, run alone (their counters already satisfied: no producer beside them) --------
B, cs = 256, 16
Rc = cs * B                                                       # rows per published chunk
nch = R // Rc
ready = torch.full((nch,), 1 << 20, dtype=torch.int32, device=dev)     # every chunk "published"
done = torch.zeros((nch,), dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
t(lambda: ops.gemm(hs, wt, xp, R, GH, H, trans_b=True, bias=bias, c_layout=hl.TILE16, max_blocks=64, chunk_rows=Rc, chunk_wait=ready,
                   chunk_wait_value=1, chunk_done=done, chunk_status=status), f, "proj_ws_k: xp = hs W^T, weights-stationary, 64 workgroups")
Rd = 64 * 1024                                                     # decode at 1024 windows: 64-step chunks, 64 row blocks per workgroup
if R % Rd == 0:
    t(lambda: ops.gemm(hs, wt, xp, R, GH, H, trans_b=True, bias=bias, c_layout=hl.TILE16, max_blocks=64, chunk_rows=Rd, chunk_wait=ready,
                       chunk_wait_value=1, chunk_done=done, chunk_status=status), f, "proj_ws_k, 65536-row chunks (decode at 1024 windows), 64 workgroups")
t(lambda: ops.gemm(da, wc, dx, R, H, GH, trans_b=True, c_layout=hl.TILE16, max_blocks=64, chunk_rows=Rc, chunk_reverse=True, chunk_wait=ready,
                   chunk_wait_value=1, chunk_done=done, chunk_status=status), f, "dX = da W, persistent chunked gemm_fast_k, 64 workgroups")
def kstream(n_wg):
    def parts(M, N):
        tiles = -(-M // 128) * -(-N // 128)
        P = 1
        while P * 2 * tiles <= n_wg and Rc % (P * 2 * 64) == 0:
            P *= 2
        return P
    kw = dict(k_wait=ready, k_wait_value=1, k_chunk_rows=Rc, k_reverse=True, chunk_status=status, trans_a=True, accumulate=True, build_only=True)
    probs = [ops.gemm(hs, da, dU, H, GH, R, split_k=parts(H, GH), colsum_b=cs_, **kw) for cs_ in (cs1, cs2)]
    probs += [ops.gemm(hs, da, dU2, H, GH, R, split_k=parts(H, GH), **kw), ops.gemm(idx, da, dW, 61, GH, R, a_kind=hl.ONEHOT, split_k=parts(61, GH), **kw)]
    ops.gemm_kstream_multi(probs)
cs1, cs2 = torch.zeros((GH,), device=dev), torch.zeros((GH,), device=dev)
t(lambda: kstream(32), 3 * f + 2.0 * R * 61 * GH, "gemm_kstream_multi_k: dU, dW of two layers (4 problems, 32 workgroups each)")
assert int(status.item()) == 0
