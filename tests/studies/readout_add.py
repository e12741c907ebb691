#!/usr/bin/env python3
"""SURVEY A.6 / F9 hedge: what would the OTHER reading of recurrentshop's readout cost?  The decoder as written steps on a constant
input (x_t = start: the readout is accepted and dropped, vae_definition.py:532-546,570); readout='add' feeds the previous step's
output back, x_t = start + y_{t-1}.  Oracle only, forward only (oracle/vae_oracle.py cfg['readout']); excluded from every parity claim.
Prints the losses of both readings on the same parameters and batches: at initialisation, and after the as-written model has been
trained for a few hundred oracle steps (how far apart the two readings are once the weights mean something).
   python tests/studies/readout_add.py [--cell GRU] [--steps 150]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.vae_oracle import OracleVAE, make_cfg          # noqa: E402
from tests.oracle_util import tiny_problem                  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="GRU")
ap.add_argument("--steps", type=int, default=150)
a = ap.parse_args()
cfg, p, batch, eps, m = tiny_problem(a.cell, B=16, H=32, Z=16, T=16, V=4, seed=7)
batch.pop("w_notes", None)
cfg_add = dict(cfg, readout="add")
m_add = OracleVAE(make_cfg(**cfg_add))


def both(tag):
    l0, l1 = m.forward(p, batch, eps)[0], m_add.forward(p, batch, eps)[0]
    keys = [k for k in ("loss", "notes_loss", "instr_loss", "vel_loss", "notes_acc") if k in l0]
    print("%-28s " % tag + "  ".join("%s %.4f | %.4f" % (k, l0[k], l1[k]) for k in keys) + "    (as written | readout=add)")


both("at initialisation")
st = m.new_opt_state(p)
m.cfg["lr"] = 2e-3
for i in range(a.steps):
    m.train_step(p, st, batch, eps)
both("after %d steps as written" % a.steps)
