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
