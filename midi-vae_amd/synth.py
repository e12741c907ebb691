"""Synthetic piano-roll windows with the statistics of the reference's window tensor.

Layout facts reproduced (reference import_midi.py:245-286,303-345): rows interleave ``max_voices`` monophonic
voices (row i + step*V = voice i at tick step); every row is one-hot over ``high_crop-low_crop`` pitches plus a
trailing SILENT class; velocity is 0 where nothing is struck and in [0.5, 1] where a note starts
(import_midi.py:273); one 16-way GM instrument category per voice (midi_functions.py:22-27).
"""
from __future__ import annotations

import numpy as np


def make_windows(n, T, D=61, V=4, ID=16, C=2, Z=256, seed=1234, epsilon_std=0.01, p_silent=0.35):
    """Returns dict of host arrays in the engine's staging format:
    x_idx (n,T) u8, i_idx (n,V) u8, vel (n,T) f32, c_idx (n,) u8, hist (n,Z) f32 zeros, eps (n,Z) f32."""
    rng = np.random.default_rng(seed)
    x_idx = np.where(rng.random((n, T)) < p_silent, D - 1, rng.integers(0, D - 1, (n, T))).astype(np.uint8)
    struck = (x_idx != D - 1) & (rng.random((n, T)) >= 0.5)
    vel = np.where(struck, 0.5 + 0.5 * rng.random((n, T)), 0.0).astype(np.float32)
    i_idx = rng.integers(0, ID, (n, V)).astype(np.uint8)
    c_idx = rng.integers(0, max(C, 1), (n,)).astype(np.uint8)
    hist = np.zeros((n, Z), np.float32)          # epoch-0 behaviour of reference vae_training.py:790-791
    eps = (rng.standard_normal((n, Z)) * epsilon_std).astype(np.float32)
    return dict(x_idx=x_idx, i_idx=i_idx, vel=vel, c_idx=c_idx, hist=hist, eps=eps)


def to_reference_format(w, D=61, ID=16):
    """The same windows as the reference's (X, Y, C, I, V, D) NumPy arrays (one song = these n windows)."""
    n, T = w["x_idx"].shape
    X = np.zeros((n, T, D))
    np.put_along_axis(X, w["x_idx"][..., None].astype(np.int64), 1, -1)
    I = np.zeros((w["i_idx"].shape[1], ID))
    I[np.arange(I.shape[0]), w["i_idx"][0]] = 1
    return X, X.copy(), int(w["c_idx"][0]), I, w["vel"].astype(np.float64), np.zeros((n, T))


def elbo_inputs(cell, T, B, V, Z, C, seed=1234):
    """The first ``B`` windows of bench.py's rank-0 inputs (make_windows with the bench's seed) and its initial parameters: what
    both sides of bench.py's ELBO comparison start from - the engine directly, the float64 torch-CPU restatement
    (oracle/torch_cpu.py elbo_trajectory) by importing this.  Data only, no arithmetic."""
    from .layout import ModelSpec, init_params
    spec = ModelSpec(cell=cell, H=256, Z=Z, Din=61, Dout=61, T=T, V=V, ID=16, C=C, Le=2, Ld=2)
    w = make_windows(max(B, 256), T, 61, V, 16, C, Z, seed=seed, epsilon_std=spec.epsilon_std)
    w = {k: v[:B] for k, v in w.items()}
    return spec, w, init_params(spec, seed)


def elbo_epsilon(step, B, Z, epsilon_std, seed=1234):
    """the fresh draw of optimizer step ``step`` (seed s + step), already scaled"""
    return (np.random.default_rng(seed + 1 + step).standard_normal((B, Z)) * epsilon_std).astype(np.float32)
