"""ELBO of the engine against the float64 oracle over MANY optimizer steps (north_star 'ELBO within 1e-3 of
reference after equal steps').

The benched schedule - H=256 bf16, resident slot-interleaved kernels, time-pipelined stacks, K-streaming gradients, fused latent
chain - at T=512, 16 windows, a FRESH epsilon per step, Keras Adam; the oracle (oracle/vae_oracle.py) steps from the same initial
parameters on the same draws.  |engine - oracle| of the ELBO (Keras total loss) and of its parts is tabulated every ``--every``
steps, with the largest value so far and the first step at which the ELBO difference left the 1e-3 band (if it did).

    python tests/studies/elbo_long.py --cell LSTM --lr 2e-4 [--steps 200] [--dtype bf16]  >  profiles/r05_*_elbo_200.txt
Test infrastructure (imports oracle/): never imported by the product.  tests/test_engine_gpu.py runs a reduced size of it."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)

KEYS = ("loss", "notes_loss", "instr_loss", "vel_loss", "style_loss", "kl")


def run(cell, lr, steps, dtype="bf16", B=16, T=512, every=10, seed=31, out=sys.stdout):
    import midi_vae_amd  # noqa: F401
    from midi_vae_amd.engine import Engine
    from oracle.vae_oracle import OracleVAE, make_cfg
    from tests.test_engine_gpu import _problem, _stage
    spec, params, batch, raw = _problem(cell, B, seed=seed, H=256, Z=64, T=T, epsilon_std=0.1)
    spec.lr = lr
    orc = OracleVAE(make_cfg(**spec.oracle_cfg()))
    p = {k: v.astype(np.float64) for k, v in params.items()}
    st = orc.new_opt_state(p)
    eng = Engine(spec, max_batch=B, dtype=dtype)
    eng.set_params(params)
    print("# %s %s lr %g: %d windows, T=%d, %d Adam steps, fresh epsilon per step; |engine - oracle| per quantity" % (cell, dtype, lr, B, T, steps),
          file=out)
    print("# step    ELBO(oracle)   ELBO(engine)   " + "  ".join("%-10s" % k for k in KEYS) + "   max |dELBO| so far", file=out)
    worst, left_band, t0 = 0.0, None, time.time()
    rows = []
    for i in range(steps):
        eps = (np.random.default_rng(1000 + i).standard_normal((B, spec.Z)) * spec.epsilon_std).astype(np.float32)
        m_o = orc.train_step(p, st, batch, eps.astype(np.float64))
        raw["eps"] = eps
        _stage(eng, raw, B)
        eng.train_step(B)
        m = eng.metrics(B)
        d = {k: abs(m[k] - m_o[k]) for k in KEYS}
        worst = max(worst, d["loss"])
        if left_band is None and d["loss"] > 1e-3:
            left_band = i + 1
        rows.append((i + 1, m_o["loss"], m["loss"], d, worst))
        if (i + 1) % every == 0 or i == 0 or i + 1 == steps:
            print("%6d  %13.6f  %13.6f   " % (i + 1, m_o["loss"], m["loss"]) + "  ".join("%-10.2e" % d[k] for k in KEYS) +
                  "   %.2e" % worst, file=out)
            out.flush()
    eng.check_pipeline()
    got = eng.get_params()
    rel = max(float(np.linalg.norm(got[k].astype(np.float64) - p[k]) / (np.linalg.norm(p[k]) + 1e-30)) for k in p)
    print("# %s %s lr %g: max |dELBO| over %d steps %.2e; %s; largest relative L2 parameter difference after %d steps %.2e; "
          "ELBO %0.4f -> %0.4f; %.0f s"
          % (cell, dtype, lr, steps, worst, "inside the 1e-3 band at every step" if left_band is None else
             "FIRST step outside the 1e-3 band: %d" % left_band, steps, rel, rows[0][1], rows[-1][1], time.time() - t0), file=out)
    return worst, left_band, rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cell", default="LSTM")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--every", type=int, default=10)
    a = ap.parse_args()
    run(a.cell, a.lr, a.steps, a.dtype, T=a.T, every=a.every)
