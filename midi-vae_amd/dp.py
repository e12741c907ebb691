"""Data-parallel sharding of the train step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Windows are independent inside a step (the history latent is an INPUT, reference vae_training.py:788-798), so the
global minibatch is split contiguously over ranks and every rank holds a full replica of the ~3-4 M parameters.
The only exchange per step is the all-reduce of the flat f32 gradient buffer (Keras losses are batch means; with
equal shards the mean of per-rank means is the global mean, so gradients are summed and scaled by 1/world inside the
optimizer kernel).  The buffer is laid out encoder-first / decoder-last (layout.ParamLayout.dec_begin) so the decoder
bucket - complete when the decoder BPTT ends - can be reduced on a side stream while the encoder BPTT still runs.
"""
from __future__ import annotations


def shard_bounds(n, world, rank):
    """Contiguous split of n windows over ranks (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_allreduce(engine, dist, world):
    """Returns the hook ``engine.train_step(allreduce=...)`` expects: sums gradients across ranks and returns the
    scale (1/world) the optimizer applies."""
    def hook(grads):
        dist.all_reduce(grads, op=dist.ReduceOp.SUM)
        return 1.0 / world
    return hook
