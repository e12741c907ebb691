"""Data-parallel path on CPU: 2 processes, gloo backend.  Each rank computes the ORACLE gradients of its shard of the
global minibatch, the product's all-reduce hook (midi_vae_amd.dp) sums them, and rank 0 checks that sum/world equals
the full-batch gradient - the identity the GPU path relies on (Keras losses are batch means; equal shards)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import midi_vae_amd  # noqa: F401
from midi_vae_amd.dp import BucketedAllReduce, make_allreduce, shard_bounds
from tests.oracle_util import tiny_problem


def test_shard_bounds_cover_the_batch_contiguously():
    for n, world in ((256, 8), (10, 4), (7, 2), (3, 4)):
        spans = [shard_bounds(n, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, p, batch, eps, m = tiny_problem("GRU", B=8, seed=9)
        batch.pop("w_notes")                           # equal shards + unit weights: mean of shard means = global mean
        lo, hi = shard_bounds(8, world, rank)
        sub = {k: v[lo:hi] for k, v in batch.items()}
        met, c = m.forward(p, sub, eps[lo:hi])
        g = m.backward(p, c)
        names = sorted(g)
        flat = torch.from_numpy(np.concatenate([g[k].ravel() for k in names]))
        hook = make_allreduce(None, dist, world)
        scale = hook(flat)
        # the same reduction in two buckets (decoder side early, the rest after the backward pass) - engine.backward's use
        flat2 = torch.from_numpy(np.concatenate([g[k].ravel() for k in names]))
        cut = flat2.numel() // 3
        hook2 = BucketedAllReduce(dist, world, cut)
        hook2.early(flat2[cut:])
        assert hook2(flat2) == scale and hook2._work is None
        assert torch.equal(flat2, flat)
        loss = torch.tensor([met["loss"]], dtype=torch.float64)
        dist.all_reduce(loss)
        if rank == 0:
            met_f, c_f = m.forward(p, batch, eps)
            g_f = m.backward(p, c_f)
            want = np.concatenate([g_f[k].ravel() for k in names])
            out.put((float(np.abs(flat.numpy() * scale - want).max()), float(np.abs(want).max()),
                     float(loss.item() / world - met_f["loss"])))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    err, scale, dloss = out.get(timeout=10)
    assert err < 1e-12 * max(1.0, scale)
    assert abs(dloss) < 1e-12
