"""Host packers of the C ABI (csrc/hostpack.cpp) - no GPU needed: the reference's float64 one-hot windows -> one byte per row,
time-major, padded; bit-exact against NumPy, and against the golden vectors captured from the reference's own packers."""
import ctypes as C

import numpy as np
import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd import hiplib as hl
from midi_vae_amd import packers as pk
from midi_vae_amd.staging import Norm, host_onehot_to_index


def _onehot(idx, K, dtype):
    out = np.zeros(idx.shape + (K,), dtype)
    np.put_along_axis(out, idx[..., None].astype(np.int64), 1, -1)
    return out


@pytest.mark.parametrize("dtype,kind", [(np.float64, hl.HOST_F64), (np.float32, hl.HOST_F32), (np.uint8, hl.HOST_U8)])
@pytest.mark.parametrize("threads", [1, 4])
def test_onehot_windows_to_time_major_indices(dtype, kind, threads):
    lib = hl.load()
    assert lib.mvae_host_threads(threads) == threads
    rng = np.random.default_rng(3)
    n, T, K = 37, 64, 61
    idx = rng.integers(0, K, (n, T)).astype(np.uint8)
    X = _onehot(idx, K, dtype)
    for lo, hi, Bp in ((0, n, 48), (5, 21, 16), (30, 37, 16), (7, 7, 16)):
        out = np.full((T, Bp), 77, np.uint8)
        bad = C.c_int64(-1)
        rc = lib.mvae_host_onehot_to_index_tm(X.ctypes.data, kind, n, T, K, lo, hi, out.ctypes.data, Bp, 255, C.byref(bad))
        assert rc == 0 and bad.value == -1
        assert np.array_equal(out[:, :hi - lo], idx[lo:hi].T)
        assert np.all(out[:, hi - lo:] == 255)            # padding windows carry the "no target" index
    lib.mvae_host_threads(0)


def test_rows_that_are_not_one_hot_are_reported_not_converted():
    lib = hl.load()
    n, T, K = 6, 8, 5
    X = _onehot(np.zeros((n, T), np.uint8), K, np.float64)
    out = np.zeros((T, 16), np.uint8)
    bad = C.c_int64(-1)
    for (w, t, k, v) in ((4, 3, 2, 1.0), (2, 7, 0, 0.0), (5, 0, 1, 0.5), (1, 1, 4, -1.0), (3, 2, 0, np.nan)):
        Xb = X.copy()
        Xb[w, t, k] = v                                # two ones / all-zero / a fraction / a negative entry / NaN
        rc = lib.mvae_host_onehot_to_index_tm(Xb.ctypes.data, hl.HOST_F64, n, T, K, 0, n, out.ctypes.data, 16, 0, C.byref(bad))
        assert rc == hl.E_FORMAT and bad.value == w * T + t, (w, t, k, v, rc, bad.value)
        with pytest.raises(NotImplementedError):
            host_onehot_to_index(Xb)
    # a window outside [lo, hi) is not looked at
    Xb = X.copy()
    Xb[5, 0, 0] = 0.0
    assert lib.mvae_host_onehot_to_index_tm(Xb.ctypes.data, hl.HOST_F64, n, T, K, 0, 5, out.ctypes.data, 16, 0, C.byref(bad)) == 0
    # argument validation
    assert lib.mvae_host_onehot_to_index_tm(None, 0, n, T, K, 0, n, out.ctypes.data, 16, 0, None) == hl.E_ARG
    assert lib.mvae_host_onehot_to_index_tm(X.ctypes.data, 0, n, T, K, 0, n, out.ctypes.data, 4, 0, None) == hl.E_ARG   # Bp < B
    assert lib.mvae_host_onehot_to_index_tm(X.ctypes.data, 9, n, T, K, 0, n, out.ctypes.data, 16, 0, None) == hl.E_ARG


def test_index_and_value_rows_to_time_major():
    lib = hl.load()
    rng = np.random.default_rng(5)
    n, T = 23, 12
    idx = rng.integers(0, 61, (n, T)).astype(np.uint8)
    out = np.zeros((T, 16), np.uint8)
    assert lib.mvae_host_index_to_tm(idx.ctypes.data, n, T, 4, 17, out.ctypes.data, 16, 9) == 0
    assert np.array_equal(out[:, :13], idx[4:17].T) and np.all(out[:, 13:] == 9)
    for dtype, kind in ((np.float64, hl.HOST_F64), (np.float32, hl.HOST_F32)):
        V = rng.random((n, T)).astype(dtype)
        o = np.full((T, 32), 5.0, np.float32)
        assert lib.mvae_host_rows_to_tm_f32(V.ctypes.data, kind, n, T, 2, 23, 0.25, o.ctypes.data, 32) == 0
        assert np.array_equal(o[:, :21], (0.25 * V[2:23].T.astype(np.float32)).astype(np.float32))
        assert np.all(o[:, 21:] == 0)


def test_matches_the_reference_pinned_packer_vectors(golden):
    """tests/golden X was produced by the reference's own import format: the native converter agrees with the pure-NumPy
    packer helper that the golden tests pin (packers.onehot_to_index)."""
    X = golden["X"]
    assert np.array_equal(host_onehot_to_index(X), pk.onehot_to_index(X))


def test_global_normalisers_of_a_minibatch():
    """Keras weighted objective: score * w / mean(w != 0), then the mean -> sum(score * w) / count(w != 0)"""
    w = np.ones((10, 4))
    w[2:4, 1] = 0
    nm = Norm.of(2, 8, 4, w_notes=w, w_instr=np.array([1.0] * 5 + [0.0] * 5))
    assert (nm.B, nm.nz_notes, nm.nz_instr, nm.nz_vel, nm.nz_style) == (6, 22, 3, 6, 6)


def test_twohot_rows_to_two_index_rolls():
    """attach_instruments rows (pitch one-hot | instrument one-hot, reference import_midi.py:288-292): one pass gives both index
    rolls, time-major and padded; any other row is refused with the LOWEST offending flat row."""
    import ctypes as C
    lib = hl.load()
    n, T, K, K1, Bp = 7, 16, 77, 61, 16
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, K1, (n, T)), rng.integers(0, K - K1, (n, T))
    for dt, kind in ((np.float64, hl.HOST_F64), (np.float32, hl.HOST_F32), (np.uint8, hl.HOST_U8)):
        X = np.zeros((n, T, K), dt)
        np.put_along_axis(X, a[..., None], 1, -1)
        np.put_along_axis(X, (K1 + b)[..., None], 1, -1)
        o1, o2 = np.zeros((T, Bp), np.uint8), np.zeros((T, Bp), np.uint8)
        bad = C.c_int64(-1)
        assert lib.mvae_host_twohot_to_index_tm(X.ctypes.data, kind, n, T, K, K1, 2, n, o1.ctypes.data, o2.ctypes.data, Bp, 255,
                                                C.byref(bad)) == 0
        assert np.array_equal(o1[:, :n - 2].T, a[2:]) and np.array_equal(o2[:, :n - 2].T, b[2:])
        assert (o1[:, n - 2:] == 255).all() and (o2[:, n - 2:] == 255).all()
        X[5, 3, K1 + int(b[5, 3])] = 0                         # no instrument column in window 5, row 3
        X[6, 1, (int(a[6, 1]) + 1) % K1] = 1                    # two pitch columns in window 6, row 1
        assert lib.mvae_host_twohot_to_index_tm(X.ctypes.data, kind, n, T, K, K1, 0, n, o1.ctypes.data, o2.ctypes.data, Bp, 255,
                                                C.byref(bad)) == hl.E_FORMAT
        assert bad.value == 5 * T + 3


def test_default_pool_is_a_share_of_the_cores_this_process_may_use():
    """one process per GPU under data parallelism: the packer pool is sized from the affinity mask divided by LOCAL_WORLD_SIZE
    (8 ranks x 64 threads on a 256-thread host), never more than the share, never more than 8 (round 5: a pool that
    sleeps through a train step wakes slowly, and 8 warm threads already move 90 GB/s)"""
    import os
    import subprocess
    import sys
    code = "import midi_vae_amd; from midi_vae_amd import hiplib as hl; print(hl.load().mvae_host_threads(-1))"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cores = len(os.sched_getaffinity(0))

    def threads(**env):
        e = dict(os.environ)
        e.pop("LOCAL_WORLD_SIZE", None)
        e.update(env)
        return int(subprocess.check_output([sys.executable, "-c", code], env=e, cwd=root).decode().split()[-1])
    alone, eight = threads(), threads(LOCAL_WORLD_SIZE="8")
    assert 1 <= alone <= min(cores, 8)
    assert 1 <= eight <= max(min(cores // 8, 8), 1)
    assert eight <= alone
