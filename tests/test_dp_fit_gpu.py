"""Data parallelism in the PRODUCT path: ``autoencoder.fit`` shards every minibatch over the ranks (dp.shard_bounds), normalises
by the global counts and all-reduces the gradients - checked with TWO processes on the ONE GPU of the test box over the gloo
backend (RCCL refuses two ranks on one device; gloo reduces CUDA tensors through the host).  Each rank runs the ENGINE on its
shard; after every fit call the history and the parameters must equal the single-process run on the full minibatches, to f32
summation order - ragged last minibatches, an EMPTY shard, non-trivial sample weights and the optimizer state across fit calls
included.  A second test drives ``vae_training.run_epoch`` on ragged synthetic songs on two ranks: same number of optimizer
steps on every rank by construction (no collective left unmatched - the round-1 script hung here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import midi_vae_amd  # noqa: F401
from midi_vae_amd import packers as pk
from midi_vae_amd.config import build_settings, create_kwargs
from midi_vae_amd.model import VAE
from midi_vae_amd.synth import make_windows, to_reference_format

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _songs(s, lengths, seed=5):
    out = []
    for i, n in enumerate(lengths):
        w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=seed + i)
        X, Y, C, I, V, D = to_reference_format(w)
        rng = np.random.default_rng(100 + i)
        H = rng.standard_normal((n, s["latent_dim"])) * 0.1
        x, y, sw = pk.prepare_autoencoder_input_and_output_list(s, X, Y, i % 2, I, V, D, np.zeros((n, s["signature_vector_length"])), H,
                                                                return_sample_weight=True)
        sw = [np.asarray(a, np.float64).copy() for a in sw]
        sw[0][:, ::3] = 0.5                               # temporal weights that are not all ones ...
        sw[0][n // 2, :] = 0.0                            # ... and a window that carries no notes loss at all
        out.append((x, y, sw))
    return out


def _train(cfg, dp):
    s = build_settings(**cfg["settings"])
    m = VAE().create(compute_dtype=cfg["dtype"], seed=3, **create_kwargs(s))
    m.set_data_parallel(dp)
    hist = []
    for x, y, sw in _songs(s, cfg["lengths"]):
        h = m.autoencoder.fit(x, y, epochs=1, batch_size=s["batch_size"], shuffle=False, sample_weight=sw, verbose=False)
        hist.append({k: float(v[0]) for k, v in h.history.items()})
    eng = m._shared.engine
    eng.check_pipeline()
    return hist, [np.asarray(a) for a in m.autoencoder.get_weights()], int(eng.t_done.item()) + int(eng._count_pending)


def _fit_worker(rank, world, port, cfg, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MVAE_PIPELINE="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from midi_vae_amd.dp import DataParallel
        res = _train(cfg, DataParallel(dist))
        if rank == 0:
            out.put(res)
    finally:
        dist.destroy_process_group()


def _run_ranks(target, world, *args):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (out,)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


CASES = {
    # 21 = 8 + 8 + 5 (shards 4/4, 4/4, 3/2); 17 = 8 + 8 + 1 (rank 1's last shard is EMPTY); 6: one minibatch of 6 (3/3)
    "f32_small": dict(dtype="f32", lengths=[21, 17, 6],
                      settings=dict(cell_type="GRU", lstm_size=64, latent_dim=32, input_length=4, output_length=4, batch_size=8,
                                    learning_rate=1e-3)),
    # the resident H=256 bf16 kernels (chunk-per-launch schedule: two processes share one GPU here), LSTM, 40 = 32 + 8
    "bf16_h256": dict(dtype="bf16", lengths=[40, 19],
                      settings=dict(cell_type="LSTM", lstm_size=256, latent_dim=64, input_length=16, output_length=16, batch_size=32,
                                    learning_rate=1e-3)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_two_rank_fit_equals_single_process_fit(case):
    cfg = CASES[case]
    hist2, w2, steps2 = _run_ranks(_fit_worker, 2, cfg)
    os.environ["MVAE_PIPELINE"] = "0"
    try:
        hist1, w1, steps1 = _train(cfg, None)
    finally:
        os.environ.pop("MVAE_PIPELINE")
    assert steps1 == steps2 == sum((n + cfg["settings"]["batch_size"] - 1) // cfg["settings"]["batch_size"] for n in cfg["lengths"])
    tol = 2e-5 if cfg["dtype"] == "f32" else 2e-3
    for h1, h2 in zip(hist1, hist2):
        assert set(h1) == set(h2)
        for k in h1:
            assert abs(h1[k] - h2[k]) <= tol * (1 + abs(h1[k])), (k, h1[k], h2[k])
    for a, b in zip(w1, w2):
        # Adam divides by sqrt(v): a last-bit difference in a tiny gradient moves the first steps by up to lr; compare in units of lr
        assert np.max(np.abs(a - b)) <= (3e-5 if cfg["dtype"] == "f32" else 2e-4), float(np.max(np.abs(a - b)))


def _epoch_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MVAE_PIPELINE="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import settings
        import vae_training
        from midi_vae_amd.dp import DataParallel
        settings.batch_size = 16
        s = vars(settings)
        m = VAE().create(compute_dtype="bf16", seed=1, **create_kwargs(s))
        m.set_data_parallel(DataParallel(dist))
        songs = vae_training.synthetic_songs(3, s, seed=11)              # ragged: 4 .. 31 windows each, the same on both ranks
        tr0 = vae_training.run_epoch(m, songs, s, 0, train=True)
        tr1 = vae_training.run_epoch(m, songs, s, 1, train=True)        # epoch 1: history pre-pass on the device
        eng = m._shared.engine
        eng.check_pipeline()
        steps = int(eng.t_done.item()) + int(eng._count_pending)
        w = torch.from_numpy(np.concatenate([np.asarray(a).ravel() for a in m.autoencoder.get_weights()]))
        both = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(both, w)
        if rank == 0:
            out.put((tr0, tr1, steps, [sg["X"].shape[0] for sg in songs], float((both[0] - both[1]).abs().max())))
    finally:
        dist.destroy_process_group()


def test_two_rank_training_script_epoch_runs_in_step():
    tr0, tr1, steps, lengths, drift = _run_ranks(_epoch_worker, 2)
    assert steps == 2 * sum((n + 15) // 16 for n in lengths), (steps, lengths)
    assert drift == 0.0                         # replicas stay bit-identical: same summed gradients, same update
    for tr in (tr0, tr1):
        assert np.isfinite(tr["loss"]) and np.isfinite(tr["kl_loss"]) and tr["kl_loss"] >= 0
    assert tr1["loss"] < tr0["loss"]


def _prepass_eval(cfg, dp):
    """history pre-pass (encoder.predict on the device) + evaluate of two ragged songs, as vae_training.run_epoch(train=False)
    drives them (reference vae_training.py:286-300)"""
    s = build_settings(**cfg["settings"])
    m = VAE().create(compute_dtype=cfg["dtype"], seed=3, **create_kwargs(s))
    m.set_data_parallel(dp)
    res = []
    for i, n in enumerate(cfg["lengths"]):
        w = make_windows(n, s["output_length"], s["output_dim"], s["max_voices"], 16, s["num_classes"], s["latent_dim"], seed=50 + i)
        X, Y, C, I, V, D = to_reference_format(w)
        lat = m.encoder.predict(pk.prepare_encoder_input_list(s, X, I, V, D), batch_size=s["batch_size"], device=True)
        x, y = pk.prepare_autoencoder_input_and_output_list(s, X, Y, i % 2, I, V, D, np.zeros((n, s["signature_vector_length"])), lat)
        ev = m.autoencoder.evaluate(x, y, batch_size=s["batch_size"], verbose=False)
        zhost = m.encoder.predict(pk.prepare_encoder_input_list(s, X, I, V, D), batch_size=s["batch_size"])
        res.append((lat.latent(), [float(v) for v in ev], zhost))
    m._shared.infer.check_pipeline()
    return res


def _prepass_worker(rank, world, port, cfg, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MVAE_PIPELINE="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from midi_vae_amd.dp import DataParallel
        res = _prepass_eval(cfg, DataParallel(dist))
        if rank == 1:                   # (the rank with the SMALLER shares: it must hold every window's z and the global metrics too)
            out.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(CASES))
def test_two_rank_sharded_prepass_and_evaluate_equal_single_process(case):
    """``encoder.predict`` and ``autoencoder.evaluate`` shard each song's windows over the ranks (the pre-pass
    rows are exchanged, the metric accumulators summed): every rank ends up with the single-process z of EVERY window and the
    single-process metrics - odd window counts (21 = 11 + 10, 17 = 9 + 8) and a song shorter than two shards of 16 included."""
    cfg = dict(CASES[case], lengths=[21, 17, 3] if CASES[case]["dtype"] == "f32" else [40, 19])
    got = _run_ranks(_prepass_worker, 2, cfg)
    os.environ["MVAE_PIPELINE"] = "0"
    try:
        want = _prepass_eval(cfg, None)
    finally:
        os.environ.pop("MVAE_PIPELINE")
    tol = 2e-5 if cfg["dtype"] == "f32" else 2e-3
    for (z2, e2, h2), (z1, e1, h1) in zip(got, want):
        np.testing.assert_allclose(z2, z1, rtol=0, atol=1e-5)
        np.testing.assert_allclose(h2, h1, rtol=0, atol=1e-5)
        for a, b in zip(e2, e1):
            assert abs(a - b) <= tol * (1 + abs(b)), (e2, e1)
