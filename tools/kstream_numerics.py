"""dU of the bottom encoder layer: K-streaming launch (pipelined step) and split-K launches (chunk-per-launch schedule) against a
float64 product of the SAME saved device buffers (hs, da) - which of the two orderings is closer, and by how much they differ."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd import ops, hiplib as hl
from test_engine_gpu import _problem, _stage, _rel_l2

for cell, B, T, chunk in (("GRU", 32, 64, 8), ("LSTM", 32, 64, 8), ("GRU", 64, 128, 16)):
    spec, params, batch, raw = _problem(cell, B, seed=42, H=256, Z=64, T=T)
    res = {}
    for pipe in (True, False):
        eng = Engine(spec, max_batch=B, dtype="bf16")
        eng.pipeline, eng.pipe_chunk = pipe, chunk
        eng.set_params(params); _stage(eng, raw, B); eng.forward_backward(B)
        torch.cuda.synchronize()
        g = eng.get_grads()
        p = "enc.notes.0"
        hs = eng._v(p + ".hs", T + 1, B, spec.H); da = eng._v(p + ".da", T, B, spec.GH)
        # saved sequences are TILE16 images: bring them to row-major through the engine's own relayout
        def rowmajor(t, rows, cols):
            out = torch.empty_like(t); ops.relayout(t.contiguous(), out, rows, cols, False); return out
        hsr = hs[:T].reshape(T * B, spec.H); dar = da.reshape(T * B, spec.GH)
        if eng._seq_layout(eng.enc_notes[0]) != hl.ROWMAJOR:
            pass        # hs / da are kept row-major for the GEMMs (the TILE16 images are separate buffers)
        ref = hsr.double().cpu().numpy().T @ dar.double().cpu().numpy()
        if cell == "GRU":       # candidate block uses r*h
            rh = eng._v(p + ".rh", T, B, spec.H).reshape(T * B, spec.H)
            ref[:, 2 * spec.H:] = rh.double().cpu().numpy().T @ dar.double().cpu().numpy()[:, 2 * spec.H:]
        res[pipe] = (g[p + ".U"], ref)
    (g1, r1), (g0, r0) = res[True], res[False]
    print(cell, B, T, chunk, "kstream vs f64 %.2e   split-K vs f64 %.2e   kstream vs split-K %.2e   refs differ %.2e" %
          (_rel_l2(g1, r1), _rel_l2(g0, r0), _rel_l2(g1, g0), _rel_l2(r1, r0)))
