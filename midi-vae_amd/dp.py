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
    gradient: the gradient hook (``hook(engine)``) sums the flat f32 gradient buffer (every rank, every minibatch - also ranks
    whose shard is empty) and returns the scale 1.0; ``allreduce_sum`` sums the loss / metric accumulators once per fit /
    evaluate call and exchanges the rows of a sharded encoder pre-pass; ``allreduce_max`` keeps the pipeline status word equal
    on all ranks.  ``dist`` is torch.distributed (backend "nccl" = RCCL over xGMI on a node of MI355X; "gloo" in the CPU /
    single-GPU tests).

    Overlap policy (``overlap=None``): with RCCL and more than one rank the decoder-side gradient bucket - complete when the
    decoder BPTT ends, ~60 % of the buffer - is reduced on a communication stream BESIDE the encoder BPTT and only the encoder
    bucket is left for after the backward pass (BucketedAllReduce; north_star: all-reduce overlapped with the backward
    recurrence).  With gloo (CPU collectives on host copies) or one rank it is one all-reduce after the backward pass - through
    a ONE-rank RCCL group the hand-over costs more than there is to hide, which says nothing about 8
    ranks over xGMI.  MVAE_DP_OVERLAP=0/1 overrides."""

    def __init__(self, dist, group=None, overlap=None):
        import os
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if overlap is None:
            env = os.environ.get("MVAE_DP_OVERLAP")
            overlap = (env == "1") if env in ("0", "1") else (self.world > 1 and str(dist.get_backend(group)) == "nccl")
        self.overlap = bool(overlap)
        self._hooks = {}

    def hook(self, engine):
        """the gradient hook of ``engine.train_step*(allreduce=...)``: SUM over the ranks, optimizer scale 1"""
        h = self._hooks.get(id(engine))
        if h is None or h[0] is not engine:
            h = (engine, BucketedAllReduce(self.dist, self.world, engine.layout.dec_begin, overlap=self.overlap, group=self.group,
                                           scale=1.0))
            self._hooks = {id(engine): h}
        return h[1]

    def allreduce_grads(self, grads):
        self.dist.all_reduce(grads, op=self.dist.ReduceOp.SUM, group=self.group)
        return 1.0

    def allreduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def allreduce_max(self, t):
        """the engine's pipeline status word behind every gradient all-reduce: a rank whose step timed out makes EVERY rank skip
        that update (Engine.optimizer_step) - replicas never diverge, and every rank raises at the same metric read"""
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)


class BucketedAllReduce:
    """The hook ``engine.train_step(allreduce=...)`` expects, in two buckets.

    ``early(bucket)`` is called by the engine in the middle of the backward pass - on a communication stream that already
    waits for every producer of the decoder-side gradients [dec_begin, total) - and starts their all-reduce while the
    encoder BPTT still runs.  Calling the hook with the whole buffer after the backward pass reduces what is left
    ([0, dec_begin), or everything if ``early`` was not reached), makes the current stream wait for the early bucket and returns
    the scale the optimizer applies (``scale``; default 1/world: the mean of per-rank gradients of bench-style callers whose ranks
    train on their OWN minibatches)."""

    def __init__(self, dist, world, dec_begin, overlap=True, group=None, scale=None):
        self.dist, self.world, self.dec_begin, self.group = dist, world, int(dec_begin), group
        self.overlap = bool(overlap) and 0 < self.dec_begin
        self.scale = (1.0 / world) if scale is None else float(scale)
        self._work = None
        self.timing = None          # a list: (tag, start event, end event) per collective, on the stream it was issued from (bench.py)

    def _timed(self, tag, fn):
        if self.timing is None:
            return fn()
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.timing.append((tag, e0, e1))
        return out

    def early(self, bucket):
        if self.overlap:
            self._work = self._timed("early", lambda: self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group,
                                                                           async_op=True))

    def __call__(self, grads):
        if self._work is None:
            self._timed("late", lambda: self.dist.all_reduce(grads, op=self.dist.ReduceOp.SUM, group=self.group))
        else:
            self._timed("late", lambda: self.dist.all_reduce(grads[:self.dec_begin], op=self.dist.ReduceOp.SUM, group=self.group))
            self._work.wait()           # device-side: the current stream waits for the collective's stream
            self._work = None
        return self.scale


def make_allreduce(engine, dist, world, overlap=False):
    """Returns the hook ``engine.train_step(allreduce=...)`` expects for bench-style callers (every rank its own minibatch): sums
    gradients across ranks and returns the scale (1/world) the optimizer applies.  ``overlap=True`` (needs the engine): the
    decoder bucket is reduced beside the encoder BPTT."""
    dec_begin = engine.layout.dec_begin if engine is not None else 0
    return BucketedAllReduce(dist, world, dec_begin, overlap=overlap)
