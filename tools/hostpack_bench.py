#!/usr/bin/env python3
"""The host packers alone (no GPU): one 256-window minibatch of float64 one-hot rows (T=512, 61 wide = 64 MB) out of a 1024-window
song -> byte indices, by thread count and CPU affinity.   python tools/hostpack_bench.py [--threads 64] [--cpus 0-63]"""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=-1)
ap.add_argument("--cpus", default="", help="restrict the process to these CPUs first (a-b)")
a = ap.parse_args()
if a.cpus:
    lo, hi = (int(v) for v in a.cpus.split("-"))
    os.sched_setaffinity(0, range(lo, hi + 1))
import midi_vae_amd  # noqa
from midi_vae_amd import hiplib as hl
lib = hl.load()
if a.threads >= 0:
    lib.mvae_host_threads(a.threads)
n, T, K = 1024, 512, 61
rng = np.random.default_rng(0)
X = np.zeros((n, T, K))
X[np.arange(n)[:, None], np.arange(T)[None], rng.integers(0, K, (n, T))] = 1.0
out = np.empty((T, 256), np.uint8)
bad = C.c_int64(-1)
ts = []
for rep in range(40):
    lo = (rep % 4) * 256
    t0 = time.perf_counter()
    rc = lib.mvae_host_onehot_to_index_tm(X.ctypes.data, 0, n, T, K, lo, lo + 256, out.ctypes.data, 256, 0, C.byref(bad))
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0
ts = np.array(ts[4:])
print("threads %3d cpus %-8s: 64 MB minibatch in %.2f ms median (min %.2f, max %.2f) = %.0f GB/s" % (
    lib.mvae_host_threads(-1), a.cpus or "all", np.median(ts), ts.min(), ts.max(), 64e6 * 1.0 / (np.median(ts) * 1e-3) / 1e9))
