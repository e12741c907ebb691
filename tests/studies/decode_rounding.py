"""CPU study (oracle arithmetic only, no GPU): which bf16 rounding of the device's decode path moves the argmax note index?

VERDICT r04 next #1(b).  The notes head of the decoder (reference vae_definition.py:519-570: cell stack on a CONSTANT input,
Dense + softmax on the top cell) is stepped T times in float64 with the oracle's cell arithmetic (oracle/vae_oracle.py
rnn_forward) and, beside it, with each rounding the bf16 device path applies switched on ALONE and together:

  W   the weights the matrix cores read: U of both cells, W of the upper cell, the output kernel -> bf16 (round to nearest even)
  h   the activations the matrix cores read: h_{t-1} as the operand of h.U, the stored h sequence the upper cell's x.W and the
      output Dense read -> bf16 (state, gates and the f32 accumulation stay exact)
  xp  the stored x.W + b sequence of the upper cell -> bf16
  h2  'bf16x2': like h, but the operand is h_hi + h_lo with h_hi = bf16(h), h_lo = bf16(h - h_hi) (16 mantissa bits)
  f32 nothing rounded to bf16, but EVERY operation in float32 - the precision the reference's own Keras backend computes in

Weights are made DECISIVE (tests/test_golden_gpu.py recipe: untrained, the decoder relaxes to a uniform softmax and an argmax
comparison tests rounding noise): biases ~ N(0, bias_std), output kernel x out_gain, and optionally the recurrent kernels
x u_gain (> 1: a non-contractive decoder, the worst case for feedback of rounding errors).

Run:  python tests/studies/decode_rounding.py [--T 4096] [--B 16] > profiles/r05_a_decode_rounding_study.txt
Test infrastructure: imports oracle/, never imported by the product."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle.vae_oracle import hard_sigmoid, softmax  # noqa: E402


def bf16(x):
    """float64/float32 -> nearest bf16 (ties to even), returned as float64."""
    a = np.asarray(x, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def split2(x):
    hi = bf16(x)
    return hi + bf16(x - hi)


def cell_step(cell, xp_t, h, c, U, opnd):
    """One step of the oracle's recurrence (oracle/vae_oracle.py:191-205) with the matmul operand of h passed through ``opnd``."""
    H = U.shape[0]
    if cell == "GRU":
        a = xp_t[:, :2 * H] + opnd(h) @ U[:, :2 * H]
        z, r = hard_sigmoid(a[:, :H]), hard_sigmoid(a[:, H:])
        hh = np.tanh(xp_t[:, 2 * H:] + opnd(r * h) @ U[:, 2 * H:])
        return z * h + (1.0 - z) * hh, None
    a = xp_t + opnd(h) @ U
    i, f = hard_sigmoid(a[:, :H]), hard_sigmoid(a[:, H:2 * H])
    g, o = np.tanh(a[:, 2 * H:3 * H]), hard_sigmoid(a[:, 3 * H:])
    c = f * c + i * g
    return o * np.tanh(c), c


def decode_notes(cell, p, zh, T, mode):
    """mode: set of {'W','h','xp','h2'}.  Returns probs (B,T,D)."""
    if "f32" in mode:          # every operation in float32 (what the reference's own Keras backend computes in)
        p = {k: v.astype(np.float32) for k, v in p.items()}
        zh = zh.astype(np.float32)
    ident = lambda v: v
    wq = bf16 if "W" in mode else ident
    opnd = split2 if "h2" in mode else (bf16 if "h" in mode else ident)
    xq = bf16 if "xp" in mode else ident
    ns = 2 if cell == "LSTM" else 1
    st = []
    for l in range(2):
        st.append([np.tanh(zh @ p["dec.notes.init.%d.%d.W" % (l, s)] + p["dec.notes.init.%d.%d.b" % (l, s)]) for s in range(ns)])
    U0, U1 = wq(p["dec.notes.0.U"]), wq(p["dec.notes.1.U"])
    W1, Wo = wq(p["dec.notes.1.W"]), wq(p["dec.notes.out.W"])
    xp0 = np.broadcast_to(p["dec.notes.0.b"], (zh.shape[0], U0.shape[1]))      # zero start row: x.W + b = b
    h0, c0 = st[0][0], (st[0][1] if ns == 2 else None)
    h1, c1 = st[1][0], (st[1][1] if ns == 2 else None)
    B, D = zh.shape[0], Wo.shape[1]
    out = np.empty((B, T, D), zh.dtype)
    for t in range(T):
        h0, c0 = cell_step(cell, xp0, h0, c0, U0, opnd)
        xp1 = xq(opnd(h0) @ W1 + p["dec.notes.1.b"])
        h1, c1 = cell_step(cell, xp1, h1, c1, U1, opnd)
        out[:, t] = softmax(opnd(h1) @ Wo + p["dec.notes.out.b"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=4096)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--Z", type=int, default=128)
    a = ap.parse_args()
    import midi_vae_amd  # noqa: F401  (repo-root shim)
    from midi_vae_amd.layout import ModelSpec, init_params
    print("# decode rounding study: notes head, H=256, 2 cells, T=%d, %d windows, Z=%d; float64 NumPy; agreement = share of the"
          % (a.T, a.B, a.Z))
    print("# (window, step) rows whose argmax equals the exact run's; 'gap' = the exact run's top-2 probability gap")
    modes = [("f32 arithmetic (no bf16 at all)", {"f32"}), ("W only", {"W"}), ("h only", {"h"}), ("xp only", {"xp"}), ("W+h+xp (the bf16 path)", {"W", "h", "xp"}),
             ("h2 only (split activations)", {"h2"}), ("W+h2, xp f32 (bf16x2 path)", {"W", "h2"})]
    for cell in ("LSTM", "GRU"):
        for bias_std, out_gain, u_gain in ((0.2, 8.0, 1.0), (0.5, 16.0, 1.0), (0.2, 8.0, 2.5), (0.2, 8.0, 5.0), (0.2, 8.0, 10.0)):
            spec = ModelSpec(cell=cell, H=256, Z=a.Z, Din=61, Dout=61, T=a.T, V=8, ID=16, C=4, Le=2, Ld=2)
            named = init_params(spec, 5)
            rng = np.random.default_rng(5)
            for k in named:
                if k.endswith(".b"):
                    named[k] = (rng.standard_normal(named[k].shape) * bias_std).astype(np.float32)
                if k.endswith(".out.W"):
                    named[k] = (named[k] * out_gain).astype(np.float32)
                if k.endswith(".U") and k.startswith("dec.notes"):
                    named[k] = (named[k] * u_gain).astype(np.float32)
            p = {k: v.astype(np.float64) for k, v in named.items() if k.startswith("dec.notes")}
            rz = np.random.default_rng(3)
            z = rz.standard_normal((a.B, a.Z))
            hist = np.concatenate([np.zeros((1, a.Z)), z[:-1]])
            zh = np.concatenate([z, hist], 1)
            exact = decode_notes(cell, p, zh, a.T, set())
            want = exact.argmax(-1)
            srt = np.sort(exact, -1)
            gap = srt[..., -1] - srt[..., -2]
            moving = float(np.mean(want[:, 1:] != want[:, :-1]))
            print("\n%s  bias_std %.1f  out.W x%g  U x%g : exact run: median top-2 gap %.3g, rows with gap<1e-3 %.2f %%, distinct notes %d, "
                  "argmax changes between consecutive steps %.2f %%"
                  % (cell, bias_std, out_gain, u_gain, np.median(gap), 100 * np.mean(gap < 1e-3), len(np.unique(want)), 100 * moving))
            for name, mode in modes:
                got = decode_notes(cell, p, zh, a.T, mode)
                idx = got.argmax(-1)
                differ = idx != want
                err = np.abs(got - exact)
                by = ", ".join("gap>=%g: %.3f %%" % (g, 100 * np.mean(~differ[gap >= g]) if np.any(gap >= g) else float("nan"))
                               for g in (0.0, 1e-3, 1e-2, 1e-1))
                last = ("last quarter %.3f %%" % (100 * np.mean(~differ[:, -a.T // 4:])))
                print("  %-32s agreement %s; %s; |p-p_exact| mean %.2e max %.2e" % (name, by, last, err.mean(), err.max()))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
