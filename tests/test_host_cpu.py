"""Host-side pieces of the product that need no GPU (model.EpsilonStream)."""


def test_epsilon_drawn_ahead_is_the_stream_of_successive_draws():
    """model.EpsilonStream (round 5): blocks drawn ahead on a worker thread and handed out in call order are the numbers a loop of
    per-minibatch draws takes from an equal generator (what the oracle trajectory tests do; reference: one K.random_normal per batch,
    vae_definition.py:498-502)."""
    import numpy as np
    from midi_vae_amd.model import EpsilonStream
    Z, std = 24, 0.7
    st = EpsilonStream(np.random.default_rng(11), Z, std)
    ref = np.random.default_rng(11)
    for n in (5, 0, 1, 300, 16, 2000, 7, 1024, 3):          # (crosses block boundaries, larger than a block, empty)
        got = st.take(n)
        want = (ref.standard_normal((n, Z)) * std).astype(np.float32)
        assert got.shape == (n, Z) and got.dtype == np.float32
        np.testing.assert_array_equal(got, want)


def test_kstream_partitions_give_every_workgroup_the_same_share_of_a_chunk():
    """engine_grads.kstream_parts (round 6): the K-streaming launch's GEMMs differ in tile count (GRU: dU_zr 8, dU_h 4, dense dW 12,
    one-hot dW 6; LSTM: 16, 16, 8) - with kstream_rows every workgroup gets the same k rows of a chunk, the launch's total stays what
    kstream_wgs per GEMM gave, one GEMM never gets more than 1.5 x kstream_wgs workgroups, and a partition is whole 64-row k tiles"""
    from midi_vae_amd.engine_grads import kstream_parts as parts
    rows = 16 * 256                                             # pipe_chunk x 256 windows
    gru = {t: parts(t, rows, 2048, 16) for t in (8, 4, 12, 6)}
    assert gru == {8: 2, 4: 2, 12: 2, 6: 2}
    assert sum(t * p for t, p in gru.items() if t != 6) == 3 * 16          # a dense layer's three GEMMs: 16 + 8 + 24 workgroups
    old = {t: parts(t, rows, 0, 16) for t in (8, 4, 12, 6)}
    assert old == {8: 2, 4: 4, 12: 1, 6: 2}                     # the round-2 rule: the 12-tile GEMM a whole chunk per workgroup
    assert {t: parts(t, rows, 2048, 32) for t in (16, 8)} == {16: 2, 8: 2}
    assert parts(12, 16 * 64, 2048, 16) == 1 and parts(4, 16 * 64, 2048, 16) == 1          # 64 windows: a chunk is 1024 rows
    assert parts(16, 32 * 256, 2048, 32) == 2                   # a longer chunk: 4 would be 64 workgroups > 1.5 x 32
    assert parts(4, 32 * 256, 2048, 16) == 4 and parts(4, 16 * 200, 2048, 16) == 1         # (3200 rows: halves are not whole k tiles)
