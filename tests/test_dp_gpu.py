"""Data-parallel hook on the GPU: a one-rank RCCL group (torch.distributed backend "nccl") - all a 1-GPU box can hold -
drives the engine's bucketed gradient all-reduce (decoder bucket started on the communication stream beside the encoder
BPTT, midi_vae_amd.dp.BucketedAllReduce) and must leave the same losses as the step without a hook."""
import os
import socket

import pytest
import torch

import midi_vae_amd  # noqa: F401
from midi_vae_amd.dp import make_allreduce
from midi_vae_amd.engine import Engine
from tests.test_engine_gpu import _problem, _stage

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_rccl():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dist.all_reduce(torch.zeros(4, device="cuda"))
    except Exception as e:          # no usable RCCL in this process: nothing of the product to check here
        pytest.skip("one-rank RCCL group unavailable: %r" % (e,))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_bucketed_allreduce_beside_the_encoder_bptt(one_rank_rccl, cell):
    dist = one_rank_rccl
    B = 32
    spec, params, batch, raw = _problem(cell, B, seed=43, H=256, Z=64, T=64)
    losses = {}
    for mode in ("none", "bucketed", "single"):
        eng = Engine(spec, max_batch=B, dtype="bf16")
        eng.defer_grads_rows = 0        # (T = 64 would collect the weight-gradient GEMMs behind the last recurrence: no early bucket then)
        eng.set_params(params)
        _stage(eng, raw, B)
        hook = None if mode == "none" else make_allreduce(eng, dist, 1, overlap=(mode == "bucketed"))
        out = []
        for _ in range(3):
            eng.train_step(B, allreduce=hook)
            out.append(eng.metrics(B)["loss"])
        eng.check_pipeline()
        if mode == "bucketed":
            assert 0 < eng.layout.dec_begin < eng.layout.total and eng.s_comm is not None and hook._work is None
        losses[mode] = out
    for mode in ("bucketed", "single"):
        for a, b in zip(losses[mode], losses["none"]):
            assert abs(a - b) <= 2e-3 * (1 + abs(b)), losses
    assert losses["bucketed"][2] < losses["bucketed"][0]


class _ZeroingDist(object):
    """stands in for torch.distributed: its 'all-reduce' ZEROES the tensor, in stream order.  Whatever is added to a bucket after
    its collective was issued survives in the gradient buffer - and moves the parameters."""
    class ReduceOp(object):
        SUM = "sum"

    class _Work(object):
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def __init__(self):
        self.calls = []

    def all_reduce(self, t, op=None, group=None, async_op=False):
        self.calls.append(int(t.numel()))
        t.zero_()
        ev = torch.cuda.Event()
        ev.record()
        return self._Work(ev) if async_op else None


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
def test_nothing_is_added_to_the_decoder_bucket_after_its_collective_was_issued(cell):
    """dp.BucketedAllReduce.early reduces [dec_begin, total) beside the encoder BPTT: EVERY decoder-side gradient must have been
    queued in front of it.  (Round 5 found that the schedule held the bottom decoder layer's - and now the side heads' - gradient
    GEMMs back behind the latent chain, i.e. behind the early collective: invisible with one rank, where a sum over ranks changes
    nothing.)  Here the 'collective' zeroes its bucket: with all gradients zero Adam's first moment must come out of the step as
    exactly beta_1 times what it was."""
    from midi_vae_amd.dp import BucketedAllReduce
    B = 32
    spec, params, batch, raw = _problem(cell, B, seed=53, H=256, Z=64, T=64)
    eng = Engine(spec, max_batch=B, dtype="bf16")
    eng.defer_grads_rows = 0
    eng.set_params(params)
    _stage(eng, raw, B)
    for _ in range(2):                  # (the first pipelined step of an engine is verified, and takes no early bucket)
        eng.train_step(B)
    eng.check_pipeline()
    before = eng.opt_m.clone()
    fake = _ZeroingDist()
    hook = BucketedAllReduce(fake, 2, eng.layout.dec_begin, overlap=True, scale=1.0)
    eng.train_step(B, allreduce=hook)
    eng.check_pipeline()
    torch.cuda.synchronize()
    assert fake.calls == [eng.layout.total - eng.layout.dec_begin, eng.layout.dec_begin], fake.calls
    want = before * 0.9              # (Keras Adam, beta_1 = 0.9: m <- beta_1 m + (1 - beta_1) g with g = 0)
    moved = ((eng.opt_m - want).abs() > 1e-5 * want.abs() + 1e-12).nonzero().flatten()
    import numpy as np
    names = sorted({n for n, e in eng.layout.entries.items() for i in moved[::max(1, len(moved) // 64)].tolist()
                    if e.offset <= i < e.offset + int(np.prod(e.shape))})
    assert len(moved) == 0, (len(moved), names)


@pytest.mark.parametrize("overlap", [0, 1])
def test_bench_dp_code_path_on_two_ranks_keeps_the_replicas_identical(overlap, tmp_path):
    """bench.py's own data-parallel path - make_allreduce / BucketedAllReduce with and without the early decoder bucket, the
    per-rank statistics, the collective timings - as the driver launches it (python -m torch.distributed.run, one process per
    rank), on TWO ranks sharing the one GPU of the test box over gloo, at a tiny shape: every rank trains on its own windows,
    the gradients are averaged, so after the run the parameters of the two replicas must be IDENTICAL.
    MVAE_PIPELINE=0: two processes on one GPU cannot keep a time-pipelined stack's kernels resident together."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MVAE_PIPELINE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--hidden", "64", "--seq-len", "8", "--voices", "2", "--latent", "16", "--batch", "16", "--prewarm-max", "0",
           "--no-cpu-baseline", "--dp-overlap", str(overlap)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=root, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["config"]["global_batch"] == 32
    dp = line["dp"]
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and dp["overlap"] == bool(overlap)
    assert dp["replicas_max_abs_diff"] == 0.0, dp
    assert dp["rank_median_ms_per_step"]["min"] <= dp["rank_median_ms_per_step"]["max"]
    assert dp["allreduce_ms"]["late"] is not None and (dp["allreduce_ms"]["early_decoder_bucket"] is not None) == bool(overlap)
    assert dp["host_ms_per_step"] > 0 and "replayed" in dp["plan"]
    import math
    assert math.isfinite(line["elbo"]["loss_final"])


@pytest.mark.parametrize("fail_overlap", [False, True])
def test_bench_probes_both_gradient_exchange_policies_and_survives_a_failing_one(fail_overlap):
    """with N > 1 ranks and no --dp-overlap, bench.py runs a short untimed region per policy (one all-reduce behind
    the backward pass / the decoder bucket beside the encoder BPTT), records both in dp.policy_probe and times the K steps with the
    faster one; a region that raises (forced here) loses and the line is printed all the same.  Two gloo ranks on the one GPU."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MVAE_PIPELINE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if fail_overlap:
        env["MVAE_BENCH_FAIL_OVERLAP"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--hidden", "64", "--seq-len", "8", "--voices", "2", "--latent", "16", "--batch", "16", "--prewarm-max", "0",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=root, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    probe = line["dp"]["policy_probe"]
    assert probe["late"]["ms_per_step"] > 0 and probe["late"]["error"] is None
    if fail_overlap:
        assert probe["early_bucket"]["ms_per_step"] is None and "forced failure" in probe["early_bucket"]["error"]
        assert probe["chosen"] == "late" and line["dp"]["overlap"] is False
    else:
        assert probe["early_bucket"]["ms_per_step"] > 0
        assert line["dp"]["overlap"] == (probe["chosen"] == "early_bucket")
    assert line["dp"]["replicas_max_abs_diff"] == 0.0 and line["steps"] == 3


@pytest.mark.parametrize("overlap", [False, True])
def test_data_parallel_steps_replay_as_plans_with_the_collectives_between_their_ranges(one_rank_rccl, overlap):
    """a train step with a gradient hook is recorded like any other; its collectives are HOST actions between
    the call ranges of the plan (engine_plan._host_call / plan.StepPlan.run_ranges).  Twelve steps with the bucketed hook on the
    pipelined bf16 schedule: steps replay (plan_stats), every collective is issued every step (the hook's timing list counts them),
    and losses and parameters equal those of an engine that enqueues every step from Python (use_plans off)."""
    dist = one_rank_rccl
    B, steps = 32, 12
    spec, params, batch, raw = _problem("LSTM", B, seed=47, H=256, Z=64, T=64)
    res = {}
    for plans in (True, False):
        eng = Engine(spec, max_batch=B, dtype="bf16")
        eng.use_plans = plans
        eng.defer_grads_rows = 0        # (the early bucket needs the decoder's gradients complete before the encoder BPTT)
        eng.set_params(params)
        _stage(eng, raw, B)
        hook = make_allreduce(eng, dist, 1, overlap=overlap)
        hook.timing = []
        losses = []
        for _ in range(steps):
            eng.train_step(B, allreduce=hook)
            losses.append(eng.metrics(B)["loss"])
        eng.check_pipeline()
        tags = [t for t, _, _ in hook.timing]
        assert tags.count("late") == steps and tags.count("early") == (steps - 1 if overlap else 0), tags   # (no early bucket on the unverified first step)
        res[plans] = (losses, eng.get_params(), dict(eng.plan_stats))
    assert res[True][2]["replayed"] >= steps - 8, res[True][2]       # (the first, unverified steps read the status word: never armed)
    assert res[False][2]["replayed"] == 0
    # (the same launch list; the split-K gradient sums are f32 atomics, so two runs agree to their order of summation, not bit for bit)
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 2e-4 * (1 + abs(b)), (res[True][0], res[False][0])
    import numpy as np
    for k, v in res[False][1].items():
        assert np.linalg.norm(res[True][1][k] - v) <= 2e-3 * (np.linalg.norm(v) + 1e-6) + 1e-6, k
