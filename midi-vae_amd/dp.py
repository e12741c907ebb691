"""Data-parallel sharding of the train step: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Windows are independent inside a step (the history latent is an INPUT, reference vae_training.py:788-798), so the
global minibatch is split contiguously over ranks and every rank holds a full replica of the ~3-4 M parameters.
The only exchange per step is the all-reduce of the flat f32 gradient buffer (Keras losses are batch means; with
equal shards the mean of per-rank means is the global mean, so gradients are summed and scaled by 1/world inside the
optimizer kernel).  The buffer is laid out encoder-first / decoder-last (layout.ParamLayout.dec_begin) so the decoder
bucket - complete when the decoder BPTT ends - can be reduced on a side stream while the encoder BPTT still runs
(BucketedAllReduce; optional, see make_allreduce for why one all-reduce after the backward pass is the default).
"""
from __future__ import annotations


def shard_bounds(n, world, rank):
    """Contiguous split of n windows over ranks (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallel:
    """What ``Autoencoder.fit(dp=...)`` / ``VAE.set_data_parallel`` take: the process group the minibatches are sharded over.

    fit splits every global minibatch of <= batch_size consecutive windows contiguously over the ranks (shard_bounds) and
    normalises every loss by the GLOBAL counts (staging.Norm), so each rank's gradient is its SHARE of the single-process
    gradient: ``allreduce_grads`` sums the flat f32 gradient buffer (one collective per optimizer step, every rank, every
    minibatch - also ranks whose shard is empty) and returns the scale 1.0; ``allreduce_sum`` sums the loss / metric
    accumulators once per fit call.  ``dist`` is torch.distributed (backend "nccl" = RCCL over xGMI on a node of MI355X;
    "gloo" in the CPU / single-GPU tests)."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def allreduce_grads(self, grads):
        self.dist.all_reduce(grads, op=self.dist.ReduceOp.SUM, group=self.group)
        return 1.0

    def allreduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


class BucketedAllReduce:
    """The hook ``engine.train_step(allreduce=...)`` expects, in two buckets.

    ``early(bucket)`` is called by the engine in the middle of the backward pass - on a communication stream that already
    waits for every producer of the decoder-side gradients [dec_begin, total) - and starts their all-reduce while the
    encoder BPTT still runs.  Calling the hook with the whole buffer after the backward pass reduces what is left
    ([0, dec_begin), or everything if ``early`` was not reached, e.g. under graph replay), makes the current stream wait
    for the early bucket and returns the scale (1/world) the optimizer applies."""

    def __init__(self, dist, world, dec_begin, overlap=True):
        self.dist, self.world, self.dec_begin = dist, world, int(dec_begin)
        self.overlap = bool(overlap) and 0 < self.dec_begin
        self._work = None

    def early(self, bucket):
        if self.overlap:
            self._work = self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, async_op=True)

    def __call__(self, grads):
        if self._work is None:
            self.dist.all_reduce(grads, op=self.dist.ReduceOp.SUM)
        else:
            self.dist.all_reduce(grads[:self.dec_begin], op=self.dist.ReduceOp.SUM)
            self._work.wait()           # device-side: the current stream waits for the collective's stream
            self._work = None
        return 1.0 / self.world


def make_allreduce(engine, dist, world, overlap=False):
    """Returns the hook ``engine.train_step(allreduce=...)`` expects: sums gradients across ranks and returns the
    scale (1/world) the optimizer applies.  ``overlap=True`` (needs the engine): the decoder bucket is reduced beside the
    encoder BPTT.  Off by default - measured through a one-rank RCCL group on one MI355X the hand-over to the
    communication stream (ten cross-stream waits) and the collective's launch beside the recurrent kernels cost 0.55 ms
    of a 9.4 ms step, more than the whole 14 MB all-reduce is expected to take over xGMI; one all-reduce after the
    backward pass costs 0.07 ms there."""
    dec_begin = engine.layout.dec_begin if engine is not None else 0
    return BucketedAllReduce(dist, world, dec_begin, overlap=overlap)
