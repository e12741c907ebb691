#!/usr/bin/env python3
"""bench.py - MIDI-VAE train-step (or decode) throughput on N MI355X (one process per GPU, RCCL over xGMI).

    python bench.py --gpus 1 --steps 20 --warmup 5                      (BASELINE configs[1], the default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config 1|2|3|4]

Workloads = BASELINE.json ``configs`` by index (``--config``; the per-GPU share under weak scaling):
  1  2-style, seq_len=128 x voices=4 (T=512 interleaved rows), z=64, 256 windows per GPU            train step   [default]
  2  4-style, seq_len=256 x voices=8 (T=2048), z=128, 512 windows per GPU (1024 on 2 GPUs)          train step
  3  the same with 4096 windows on 8 GPUs = 512 per GPU (velocity / instrument heads: on in every config here)  train step
  4  style-transfer decode: seq_len=512 x voices=8 (T=4096), z=128, 1024 windows per GPU (8192 on 8)  decoder forward + fused argmax
bf16 MFMA operands, LSTM cells (north_star; --cell GRU for the reference's shipped default), H=256, 2+2 layers, instrument +
velocity + style heads.  A train "step" = forward + all losses + backward + (gradient all-reduce) + Keras-Adam update on one
minibatch of synthetic piano-roll windows already resident in HBM; a decode "step" = one batch of latents through the decoder,
one byte per row (the argmax note index) written.  Weak scaling: the per-GPU batch is fixed.

Multi-GPU driver commands (what the DP configs of BASELINE.json name):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --config 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --config 3
    ... --gpus 8 --config 4        (decode: replicas, no collective)
With N > 1 ranks and no --dp-overlap the step's gradient exchange is PROBED first, untimed: a short region with one all-reduce
after the backward pass, then one with the decoder-side bucket reduced beside the encoder BPTT (dp.BucketedAllReduce).  The faster
policy runs the timed K steps; a probe that raises or whose pipeline times out is recorded and loses.  Both figures are in ``dp``.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline     dominant kernel (the T-step recurrent kernels) measured live with HIP events on the launch stream
  cpu_baseline the same step as float32 torch-CPU tensor operations on the host cores (oracle/torch_cpu.py), rank 0 / N=1 only,
               bounded sample, timed BEFORE the GPU phase
  fit_e2e      (N=1, the default run) the PRODUCT path a reference user calls: one 1024-window song per autoencoder.fit call on
               float64 one-hot lists at configs[1] - ms per optimizer step and windows/s including host conversion and upload
               (tools/fit_e2e_bench.py --json in a process of its own, after the headline region)
  elbo         (train configs, rank 0 / N=1) the ELBO after each of K_e optimizer steps with a fresh epsilon per step on the first
               16 windows of the bench's inputs: this engine (bf16, the timed schedule) beside float64 torch-CPU arithmetic
               (oracle/torch_cpu.py elbo_trajectory, in the CPU subprocess phase), and their largest difference
"""
import argparse
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def cpu_baseline(spec, B, budget_s=15.0, hard_limit_s=150.0):
    """The identical train step (forward + losses + analytic backward + Keras-Adam) as float32 torch-CPU tensor operations on the
    host cores (oracle/torch_cpu.py, pinned to the NumPy oracle by tests/test_torch_port_cpu.py) - the reference's own Keras CPU
    path cannot run here.  Full batch (B windows), bounded in TIME: the step is timed on the first T_s of the T time
    steps and scaled by T / T_s (the step's cost is linear in T), T_s chosen to fit ``budget_s``; the thread count is calibrated
    (all cores is not the fastest for ~10^5 small tensor operations per step).  Runs in a SUBPROCESS with a hard wall-clock limit,
    BEFORE the GPU phase."""
    import subprocess
    base = [sys.executable, os.path.join(ROOT, "oracle", "torch_cpu.py"), "--cell", spec.cell, "--T", str(spec.T), "--B", str(B),
            "--V", str(spec.V), "--Z", str(spec.Z), "--C", str(spec.C), "--budget", str(budget_s)]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    # One subprocess per thread count, each with its own hard limit: "all visible cores" can be pathological for ~10^5 small
    # tensor operations per step (a 256-thread OpenMP team on the 64-core EPYC of an MI355X box did not finish a step in 15
    # minutes), so the count is searched upwards from 16 and the best FINISHED attempt is reported, with the threads it used.
    best, tried = None, []
    for n, limit in ((16, hard_limit_s), (64, hard_limit_s / 2), (cores, hard_limit_s / 3)):
        if n > cores or any(n == t for t, _ in tried):
            continue
        try:
            out = subprocess.run(base + ["--threads", str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, cwd=ROOT)
            rec = json.loads(out.stdout.decode().strip().splitlines()[-1])
            tried.append((n, rec["value"]))
            if best is None or rec["value"] > best["value"]:
                best = rec
            else:
                break                                  # slower with more threads: even more will not be faster
        except (subprocess.TimeoutExpired, ValueError, IndexError):
            tried.append((n, None))
            break                                      # more threads will not finish either
    if best is None:
        best = {"value": None, "unit": "windows/s", "cores": 0, "kind": "port",
                "sample": "oracle/torch_cpu.py did not finish within %.0f s on this host" % hard_limit_s}
    best["cpu"], best["host_cores"] = model, cores
    best["threads_tried"] = [{"threads": n, "windows_per_s": v} for n, v in tried]
    return best


def cpu_elbo(spec, B, steps, limit_s=150.0):
    """ELBO trajectory of ``steps`` real optimizer steps on the first B windows of the bench's inputs in float64 torch-CPU
    arithmetic (oracle/torch_cpu.py elbo_trajectory) - a subprocess BEFORE the GPU phase, with a hard limit"""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "torch_cpu.py"), "--cell", spec.cell, "--T", str(spec.T), "--B", str(B),
           "--V", str(spec.V), "--Z", str(spec.Z), "--C", str(spec.C), "--elbo-steps", str(steps), "--threads", "16"]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s, cwd=ROOT)
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except (subprocess.TimeoutExpired, ValueError, IndexError):
        return None


def gpu_elbo(spec, B, steps, dtype, device):
    """the same trajectory on the engine: same windows, same initial parameters, same draws (synth.elbo_inputs / elbo_epsilon:
    data generators both sides import - this GPU leg imports nothing from oracle/)"""
    import torch
    from midi_vae_amd.engine import Engine
    from midi_vae_amd.synth import elbo_epsilon, elbo_inputs
    _, w, params = elbo_inputs(spec.cell, spec.T, B, spec.V, spec.Z, spec.C)
    eng = Engine(spec, max_batch=B, dtype=dtype, device=device, seed=1234)
    eng.set_params(params)
    out = []
    for i in range(steps):
        eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], elbo_epsilon(i, B, spec.Z, spec.epsilon_std))
        eng.stage_decoder_inputs(B, hist=w["hist"])
        eng.stage_targets(B, w["x_idx"], w["c_idx"])
        eng.train_step(B)
        m = eng.metrics(B)
        out.append({k: float(m[k]) for k in ("loss", "notes_loss", "instr_loss", "vel_loss", "style_loss", "kl")})
    eng.check_pipeline()
    del eng
    torch.cuda.empty_cache()
    return out


CONFIGS = {     # BASELINE.json configs[i]: (seq_len, voices, latent, classes, windows per GPU, mode, global batch the config names, GPUs)
    # 0 is NOT a BASELINE config: the configuration the reference ships (settings.py:108-112,140,155; models/BvM/params.txt) -
    # T = 16 x 4 = 64 rows, latent 256, batch 256, GRU - what `python vae_training.py` runs first
    0: (16, 4, 256, 2, 256, "train", 256, 1),
    1: (128, 4, 64, 2, 256, "train", 256, 1),
    2: (256, 8, 128, 4, 512, "train", 1024, 2),
    3: (256, 8, 128, 4, 512, "train", 4096, 8),
    4: (512, 8, 128, 4, 1024, "decode", 8192, 8),
}


def algorithmic_flops_per_window(spec):
    """Forward FLOPs of one window, SURVEY section 8(d) formula (mm(k,n) = 2kn; decoder input projections counted once per window:
    the decoder input is constant over the steps, F9; the one-hot x W of encoder layer 1 counted as the GEMM it replaces).
    A train step is 3x this (1 forward + 2 backward GEMMs)."""
    mm = lambda k, n: 2.0 * k * n
    H, GH, Z, T, V, D, ID = spec.H, spec.GH, spec.Z, spec.T, spec.V, spec.Dout, spec.ID
    s_ = spec.nstate
    f_enc = (T * (mm(spec.Din, GH) + mm(H, GH)) + (spec.Le - 1) * T * 2 * mm(H, GH) + V * (mm(ID, GH) + mm(H, GH)) +
             T * (mm(1, GH) + mm(H, GH)) + mm(3 * H, H) + mm(H, H) + 2 * mm(H // 2, Z))
    f_dec = ((spec.Ld + 2) * s_ * mm(spec.zin, H) + (mm(D, GH) + T * mm(H, GH)) + (spec.Ld - 1) * T * 2 * mm(H, GH) +
             T * mm(H, D) + (mm(ID, GH) + V * (mm(H, GH) + mm(H, ID))) + (mm(1, GH) + T * (mm(H, GH) + mm(H, 1))))
    return f_enc + f_dec


def decoder_flops_per_window(spec):
    """the decoder's share of the forward FLOPs (BASELINE configs[4]: decode only)"""
    mm = lambda k, n: 2.0 * k * n
    H, GH, T, V, D, ID = spec.H, spec.GH, spec.T, spec.V, spec.Dout, spec.ID
    return ((spec.Ld + 2) * spec.nstate * mm(spec.zin, H) + (mm(D, GH) + T * mm(H, GH)) + (spec.Ld - 1) * T * 2 * mm(H, GH) +
            T * mm(H, D) + (mm(ID, GH) + V * (mm(H, GH) + mm(H, ID))) + (mm(1, GH) + T * (mm(H, GH) + mm(H, 1))))


def workload_name(config, C, seq, voices, T, latent, B, named_batch, named_gpus, cell, decode):
    what = ("%s: %d-style seq_len=%d voices=%d (T=%d rows) z=%d batch=%d/GPU (%s) %s H=256 2+2 layers, " % (
        "BASELINE configs[%d]" % config if config else "the reference's shipped configuration (settings.py:108-112,140,155)", C, seq,
        voices, T, latent, B, "the config names %d windows on %d GPU%s" % (named_batch, named_gpus, "s" if named_gpus > 1 else ""),
        cell))
    return what + ("decoder forward on swapped latents + fused argmax decode (one byte per row leaves the chip)" if decode else
                   "notes+instrument+velocity+style heads, Keras-Adam")


def side_workload(config, cell, dtype, device, steps, warmup, step_times=False):
    """One of the OTHER workloads, measured in this process after the headline region (the driver then
    observes them): K steps bracketed by synchronize, the dominant kernel's launches of every 4th step bracketed with HIP events
    on their stream - the same measurement as the headline, shorter.  Returns the member of the line's ``other_configs`` array."""
    import numpy as np
    import torch
    from midi_vae_amd.engine import Engine
    from midi_vae_amd.layout import ModelSpec
    from midi_vae_amd.synth import make_windows
    seq, voices, latent, C, B, mode, named_batch, named_gpus = CONFIGS[config]
    T, decode = seq * voices, mode == "decode"
    spec = ModelSpec(cell=cell, H=256, Z=latent, Din=61, Dout=61, T=T, V=voices, ID=16, C=C, Le=2, Ld=2)
    eng = Engine(spec, max_batch=B, dtype=dtype, device=device, seed=1234, training=not decode)
    w = make_windows(B, T, 61, voices, 16, C, latent, seed=1234, epsilon_std=spec.epsilon_std)
    if decode:
        z = np.random.default_rng(1234).standard_normal((B, latent)).astype(np.float32)
        z[:, [0, 1]] = z[:, [1, 0]]
        eng.stage_decoder_inputs(B, z=z, hist=np.concatenate([np.zeros((1, latent), np.float32), z[:-1]]))
        step = lambda: eng.decode(B, want_probs=False)
        kinds = {("rnn_fwd", "dec.notes.1"), ("rnn_fwd_multi", "dec")}
    else:
        eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
        eng.stage_decoder_inputs(B, hist=w["hist"])
        eng.stage_targets(B, w["x_idx"], w["c_idx"])
        step = lambda: eng.train_step(B)
        kinds = {("rnn_bwd", "dec.notes.1"), ("rnn_bwd", "dec.notes.0"), ("rnn_bwd_multi", "dec")}
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.7:          # clocks up, first launches done, step plans armed
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    eng.prof_kinds = kinds
    for i in range(max(warmup, 16)):       # (both kinds of step - bracketed, plain - recorded three times and replayed once)
        eng.prof = {} if i % 4 == 0 else None
        step()
    eng.prof = None
    torch.cuda.synchronize()
    host = 0.0
    prof = {}
    per_step = []
    paced0 = eng.pace_wait_s
    t0 = time.perf_counter()
    for i in range(steps):
        eng.prof = prof if i % 4 == 0 else None        # (every 4th step is bracketed)
        h0 = time.perf_counter()
        step()
        host += time.perf_counter() - h0
        per_step.append(time.perf_counter() - h0)
    eng.prof = None
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    paced = eng.pace_wait_s - paced0                     # host time spent WAITING for the device inside step() (Engine._pace_now)
    if step_times:
        sys.stderr.write("config %d host ms per step() call: %s\n" % (config, " ".join("%.2f" % (1e3 * v) for v in per_step)))
    eng.prof = prof
    summary = eng.prof_summary()
    eng.prof = None
    eng.check_pipeline()
    dom = "rnn_fwd" if decode else "rnn_bwd"
    longk = {k: v for k, v in summary.items() if k[0] in (dom, dom + "_multi")}
    tot_ms = sum(n * ms for n, ms, _ in longk.values())
    tot_steps = sum(n * st for n, _, st in longk.values())
    Bp = (B + 15) // 16 * 16
    lpl = spec.Ld if any(k[0].endswith("_multi") for k in longk) else 1
    achieved = 2.0 * Bp * spec.H * spec.G * spec.H * tot_steps / (tot_ms * 1e-3) / 1e12 if tot_ms else float("nan")
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    ms_step = elapsed / steps * 1e3
    flop = (3.0 * algorithmic_flops_per_window(spec) if not decode else decoder_flops_per_window(spec)) * B
    out = {"workload": workload_name(config, C, seq, voices, T, latent, B, named_batch, named_gpus, cell, decode),
           "baseline_config": config, "cell": cell, "mode": mode, "steps": steps, "ms_per_step": ms_step,
           "value": B * steps / elapsed, "unit": "windows/s", "host_ms_per_step": host / steps * 1e3,
           # the paced host spends part of every step() call waiting for the device: what it WORKS is the rest
           "host_work_ms_per_step": (host - paced) / steps * 1e3, "host_paced_wait_ms_per_step": paced / steps * 1e3,
           "bound": "host" if (host - paced) > 0.9 * elapsed else "device",
           "roofline": {"kernel": dom, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                        "us_per_time_step": tot_ms * 1e3 / tot_steps * lpl if tot_steps else float("nan")},
           "whole_step": {"tflops": flop / (ms_step * 1e-3) / 1e12, "frac_of_peak": flop / (ms_step * 1e-3) / 1e12 / peak},
           "plan": _plan_block(eng)}
    del eng
    torch.cuda.empty_cache()
    return out


def _plan_block(eng, **more):
    """step-plan statistics of an engine for the line: ``refused`` = kinds of call the engine gave up making replayable, with the
    reason of each (a torch operation inside the call - the prewarm's forward_backward passes -, three recordings that differ in
    something that is not an announced counter ...); such calls are enqueued from Python every time"""
    st = eng.plan_stats
    return dict({k: v for k, v in st.items() if k != "refused"}, refused=len(st["refused"]),
                refused_kinds={str(k)[:80]: str(v)[:120] for k, v in list(st["refused"].items())[:6]}, **more)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="index into BASELINE.json configs (module docstring)")
    ap.add_argument("--cell", default=None, choices=["LSTM", "GRU"], help="default: LSTM (north_star); GRU for --config 0")
    ap.add_argument("--no-other-configs", action="store_true", help="N=1: skip the other workloads measured after the headline")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU (0 = the config's)")
    ap.add_argument("--seq-len", type=int, default=0)
    ap.add_argument("--voices", type=int, default=0)
    ap.add_argument("--latent", type=int, default=0)
    ap.add_argument("--prewarm-min", type=float, default=1.0, help="seconds of untimed passes before the warmup steps, at least")
    ap.add_argument("--prewarm-max", type=float, default=8.0, help="... at most (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--elbo-steps", type=int, default=5, help="optimizer steps of the GPU-vs-CPU ELBO comparison (0 = none)")
    ap.add_argument("--dp-overlap", type=int, default=-1,
                    help="N>1: reduce the decoder gradient bucket beside the encoder BPTT (dp.BucketedAllReduce); -1 = the product "
                         "policy (dp.DataParallel: on with RCCL and more than one rank)")
    ap.add_argument("--chunks", type=int, default=0, help="time chunks of the stacked-layer pipeline (0 = engine default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl = RCCL over xGMI; gloo: two ranks on ONE GPU in tests/test_dp_gpu.py)")
    ap.add_argument("--hidden", type=int, default=256, help="(tests) cell width")
    ap.add_argument("--step-times", action="store_true", help="(diagnostic) write every timed step's duration to stderr")
    ap.add_argument("--side", default=None, help="(internal) 'config,cell,steps,warmup': measure ONE of the other workloads in this "
                                                 "process and print its other_configs member")
    args = ap.parse_args()
    if args.side:
        cfg, cell_o, k, wu = args.side.split(",")
        import torch
        torch.cuda.set_device(0)
        print(json.dumps(side_workload(int(cfg), cell_o, args.dtype, torch.device("cuda", 0), int(k), int(wu), step_times=args.step_times)))
        return
    if args.cell is None:
        args.cell = "GRU" if args.config == 0 else "LSTM"

    import torch
    import midi_vae_amd  # noqa: F401
    from midi_vae_amd.engine import Engine
    from midi_vae_amd.layout import ModelSpec
    from midi_vae_amd.synth import make_windows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if args.backend == "gloo":
        local = 0                      # (test mode: every rank on the one GPU of the box)
    torch.cuda.set_device(local)
    dist = None
    # MVAE_BENCH_FORCE_DIST=1: go through RCCL even with one rank (checks the collective path on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("MVAE_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    seq, voices, latent, C, B, mode, named_batch, named_gpus = CONFIGS[args.config]
    seq, voices, latent, B = args.seq_len or seq, args.voices or voices, args.latent or latent, args.batch or B
    T = seq * voices
    decode = mode == "decode"
    spec = ModelSpec(cell=args.cell, H=args.hidden, Z=latent, Din=61, Dout=61, T=T, V=voices, ID=16, C=C, Le=2, Ld=2)
    device = "cuda:%d" % local
    # the CPU legs FIRST (rank 0, N=1 only): the GPU phase then runs last, undisturbed, and an idle-GPU sampler watching the
    # process sees the GPU busy at the end of the run rather than idle
    solo = world == 1 and rank == 0 and not args.no_cpu_baseline
    cpu = cpu_baseline(spec, B) if (solo and not decode) else None
    EB = 16
    elbo_cpu = cpu_elbo(spec, EB, args.elbo_steps) if (solo and not decode and args.elbo_steps > 0) else None
    elbo_gpu = gpu_elbo(spec, EB, args.elbo_steps, args.dtype, device) if elbo_cpu is not None else None
    eng = Engine(spec, max_batch=B, dtype=args.dtype, device=device, seed=1234, training=not decode)
    if args.chunks:
        eng.time_chunks = args.chunks
    w = make_windows(B, T, 61, voices, 16, C, latent, seed=1234 + rank, epsilon_std=spec.epsilon_std)
    if decode:
        import numpy as np
        z = np.random.default_rng(1234 + rank).standard_normal((B, latent)).astype(np.float32)
        z[:, [0, 1]] = z[:, [1, 0]]                 # the latent swap of the style transfer (reference vae_evaluation.py:2471-2483)
        eng.stage_decoder_inputs(B, z=z, hist=np.concatenate([np.zeros((1, latent), np.float32), z[:-1]]))
    else:
        eng.stage_encoder_inputs(w["x_idx"], w["i_idx"], w["vel"], w["eps"])
        eng.stage_decoder_inputs(B, hist=w["hist"])
        eng.stage_targets(B, w["x_idx"], w["c_idx"])

    allreduce, probe_stats = None, None
    if use_dist and not decode:
        from midi_vae_amd.dp import make_allreduce
        overlap = (world > 1) if args.dp_overlap < 0 else bool(args.dp_overlap)
        if args.dp_overlap < 0 and (world > 1 or os.environ.get("MVAE_BENCH_PROBE_DP") == "1"):
            # The first run of this path on real ranks must not be losable: both policies of the gradient
            # exchange run a short UNTIMED region each - one all-reduce behind the backward pass, then the decoder-side bucket
            # beside the encoder BPTT (the path that has only ever run on one rank) - and the faster one runs the timed steps.
            # A region that raises, or whose pipeline times out (check_pipeline: the status word is MAX-reduced over the ranks, so
            # every rank sees it), loses and is recorded; ranks agree through MAX-reduced figures.
            probe_stats, n_probe = {}, max(4, min(10, args.steps))
            for name, ov in (("late", False), ("early_bucket", True)):
                ms, err = None, None
                try:
                    if ov and os.environ.get("MVAE_BENCH_FAIL_OVERLAP") == "1":      # (tests: a forced failure still yields a line)
                        raise RuntimeError("forced failure of the early-bucket region (MVAE_BENCH_FAIL_OVERLAP=1)")
                    ar = make_allreduce(eng, dist, world, overlap=ov)
                    for _ in range(6):                        # (three recordings arm the step plan of this kind of step)
                        eng.train_step(B, allreduce=ar)
                    torch.cuda.synchronize()
                    dist.barrier()
                    tp0 = time.perf_counter()
                    for _ in range(n_probe):
                        eng.train_step(B, allreduce=ar)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - tp0) / n_probe * 1e3
                    eng.check_pipeline()
                except Exception as e:                        # noqa: BLE001 (whatever it is, the other policy still runs)
                    err = "%s: %s" % (type(e).__name__, str(e)[:200])
                t = torch.tensor([ms if err is None else 1e9, 0.0 if err is None else 1.0], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                failed = float(t[1].item()) > 0
                probe_stats[name] = {"ms_per_step": None if failed else float(t[0].item()), "steps": n_probe,
                                     "error": err if err is not None else ("failed on another rank" if failed else None)}
            lt, ea = probe_stats["late"]["ms_per_step"], probe_stats["early_bucket"]["ms_per_step"]
            overlap = ea is not None and (lt is None or ea < lt)
            probe_stats["chosen"] = "early_bucket" if overlap else "late"
        allreduce = make_allreduce(eng, dist, world, overlap=overlap)
        allreduce.timing = []          # (HIP-event pairs around the early and the late collective: dp.BucketedAllReduce)

    def step():
        if decode:
            eng.decode(B, want_probs=False)
        else:
            eng.train_step(B, allreduce=allreduce)

    # Steady state first: on a freshly started box the first GPU process runs its first second or two 10-16 % slower
    # (measured: 10.8 ms per step as the box's first process, 9.1 ms from the second process on; the recurrent kernels,
    # which live on memory latency, 5.6 instead of 3.9 us per time step) - clocks ramping up from idle.  Untimed
    # passes (no optimizer update, no collective: parameters and the reported ELBO trajectory unchanged)
    # until two consecutive blocks agree to 1 %, at least --prewarm-min and at most --prewarm-max seconds.
    if args.prewarm_max > 0:
        t_pre, prev = time.perf_counter(), None
        nblk = 20 if T <= 512 else 4
        while True:
            t0 = time.perf_counter()
            for _ in range(nblk):
                if decode:
                    eng.decode(B, want_probs=False)
                else:
                    eng.forward_backward(B)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / nblk
            el = time.perf_counter() - t_pre
            if rank == 0:
                print("prewarm: %.2f s, %.3f ms per pass" % (el, dt * 1e3), file=sys.stderr)
            if el >= args.prewarm_max or (el >= args.prewarm_min and prev is not None and abs(dt - prev) <= 0.01 * prev):
                break
            prev = dt
    # The warmup steps run exactly what the timed steps run - including the HIP-event brackets around the dominant kernel
    # (their first use costs tens of milliseconds in the first GPU process of a freshly started box: measured 12.9 instead
    # of 9.1 ms per step over 10 timed steps when the brackets first appeared inside the timed region).
    # Bracket only the dominant kernel's launches, and of those the two stacked decoder layers (plus one forward layer for the
    # critical-path figure): every event pair costs launch slots - all 26 BPTT launches bracketed slow the step by 0.5 ms.
    # (with phase launches - the engine's default - a decoder stack's two layers are ONE launch: keys (*_multi, "dec"))
    KINDS = ({("rnn_fwd", "dec.notes.1"), ("rnn_fwd_multi", "dec")} if decode else
             {("rnn_bwd", "dec.notes.1"), ("rnn_bwd", "dec.notes.0"), ("rnn_fwd", "dec.notes.1"), ("rnn_bwd_multi", "dec"),
              ("rnn_fwd_multi", "dec")})
    # The W warmup steps bracket forward AND BPTT launches (the forward figure of the critical-path bound comes from them).  Then
    # ARM untimed steps of the two kinds the timed region runs - BPTT launches bracketed / plain - four of each: three recordings
    # make a kind's enqueue a plan (engine_plan.py), the fourth replays it once.  (Round 5 found the first three plain steps of the
    # timed region Python-enqueued at 8.8 instead of 6.5 ms: the pre-warm passes run no optimizer and so arm another kind of call.)
    bwd_kinds = KINDS if decode else {k for k in KINDS if k[0].startswith("rnn_bwd")}
    ARM = 8 if args.warmup else 0
    eng.prof_kinds = KINDS
    eng.prof = {} if args.warmup else None
    for _ in range(args.warmup):
        step()
    fwd_prof = {k: v for k, v in eng.prof_summary().items() if k[0].startswith("rnn_fwd")} if args.warmup else {}
    eng.prof = {} if args.warmup else None
    for i in range(ARM):
        eng.prof_kinds = bwd_kinds if i < ARM // 2 else set()
        step()
    torch.cuda.synchronize()
    eng.prof = None
    first_loss = eng.metrics(B)["loss"] if (args.warmup and not decode) else float("nan")
    eng.prof_kinds = bwd_kinds
    eng.prof = {}               # HIP events on the launch streams (C-ABI events from a pool - Engine._timed - so a bracketed step
    #                             replays as a plan like any other)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    timed_kinds, every = eng.prof_kinds, 4       # every 4th step carries the brackets (two packets on the critical queue each)
    host_s = 0.0                # time the host spends enqueueing the steps (the step() calls themselves)
    for i in range(args.steps):
        eng.prof_kinds = timed_kinds if i % every == 0 else set()
        th = time.perf_counter()
        step()
        host_s += time.perf_counter() - th
        marks[i + 1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    raw_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    step_ms = sorted(raw_ms)
    if args.step_times:
        sys.stderr.write("headline ms per step (events): %s\n" % " ".join("%.2f" % v for v in raw_ms))
    _med = lambda v: (sorted(v)[len(v) // 2] if v else None)
    step_split = {"plain_median": _med([m for i, m in enumerate(raw_ms) if i % every]),
                  "bracketed_median": _med([m for i, m in enumerate(raw_ms) if i % every == 0]),
                  "min": step_ms[0], "max": step_ms[-1], "bracketed_every": every,
                  "slow_steps": [[i, round(m, 3)] for i, m in enumerate(raw_ms) if m > 1.05 * step_ms[len(step_ms) // 2]][:12]}
    median_ms = step_ms[len(step_ms) // 2] if args.steps % 2 else 0.5 * (step_ms[args.steps // 2 - 1] + step_ms[args.steps // 2])
    dp_stats = None
    if dist is not None:
        # per-rank step times (each rank's own event marks) and the collectives' durations: what a scaling curve is read with
        mine = torch.tensor([elapsed / args.steps * 1e3, median_ms], device="cuda", dtype=torch.float64)
        every_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_rank, mine)
        per = sorted(float(t[1].item()) for t in every_rank)
        ar = {}
        for tag in ("early", "late"):
            ms = [e0.elapsed_time(e1) for tg, e0, e1 in (allreduce.timing if allreduce is not None else []) if tg == tag]
            t = torch.tensor([sum(ms) / len(ms) if ms else -1.0], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ar[tag] = float(t.item()) if float(t.item()) >= 0 else None
        # every rank started from the same parameters and applied the same all-reduced gradients: the replicas must be identical
        pmax = eng.params.clone()
        pmin = eng.params.clone()
        dist.all_reduce(pmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(pmin, op=dist.ReduceOp.MIN)
        replica_diff = float((pmax - pmin).abs().max().item())
        dp_stats = {"rccl_ranks": dist.get_world_size(), "backend": str(dist.get_backend()), "replicas_max_abs_diff": replica_diff,
                    "rank_median_ms_per_step": {"min": per[0], "median": per[len(per) // 2], "max": per[-1]},
                    "allreduce_ms": {"early_decoder_bucket": ar["early"], "late": ar["late"],
                                     "what": "HIP-event time of the collectives on their streams, mean over the timed steps, max "
                                             "over ranks (early: the decoder-side bucket beside the encoder BPTT, null when the "
                                             "overlap is off; late: what is reduced after the backward pass)"},
                    "overlap": bool(getattr(allreduce, "overlap", False)),
                    # (a data-parallel step replays as a plan too: Python issues the collectives between its call ranges)
                    "host_ms_per_step": host_s / args.steps * 1e3, "plan": _plan_block(eng), "policy_probe": probe_stats}
        tt = torch.tensor([elapsed, median_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, median_ms = float(tt[0].item()), float(tt[1].item())
    prof = eng.prof_summary()
    for k, v in fwd_prof.items():           # (forward launches: timed over the warmup steps)
        prof.setdefault(k, v)
    eng.prof = None
    m = eng.metrics(B) if not decode else None
    eng.check_pipeline()        # raises if a time-pipelined kernel ever gave up waiting for its producer (invalid results)

    if rank == 0:
        G, H = spec.G, spec.H
        # Dominant kernel: backpropagation through time of the T-step recurrent layers (decode: their forward recurrence); every
        # bracketed launch of it in the timed region was timed with HIP events on its stream.  Algorithmic work = the recurrent
        # GEMM only: 2 * B * H * (G*H) flop per time step, summed over the steps each launch covers - SURVEY section 8d.
        dom = "rnn_fwd" if decode else "rnn_bwd"
        longk = {k: v for k, v in prof.items() if k[0] in (dom, dom + "_multi")}
        launches = sum(n for n, _, _ in longk.values())
        tot_ms = sum(n * ms for n, ms, _ in longk.values())
        tot_steps = sum(n * st for n, _, st in longk.values())
        Bp = (B + 15) // 16 * 16
        flop_step = 2.0 * Bp * H * G * H
        avg_ms = tot_ms / launches
        achieved = flop_step * tot_steps / (tot_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        # a phase launch holds BOTH layers of the stack: its work counts layer-steps, its latency per time step is launch / T
        lpl = spec.Ld if any(k[0].endswith("_multi") for k in longk) else 1
        cus = Bp // 16 * lpl                # one workgroup (= one CU) per 16 batch rows and layer
        us_dom = tot_ms * 1e3 / tot_steps * lpl
        fwdk = [v for k, v in prof.items() if k[0] in ("rnn_fwd", "rnn_fwd_multi")]
        us_fwd = sum(n * ms for n, ms, _ in fwdk) * 1e3 / max(sum(n * st for n, _, st in fwdk), 1) * lpl if fwdk else float("nan")
        ms_step = elapsed / args.steps * 1e3
        # HBM traffic of the dominant kernel from the PMC counters: bench.py cannot run a counter pass over itself, so the pass over
        # THIS command (tools/collect_profiles_r04.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, FETCH_SIZE
        # doubled as MI355X_MICROARCH.md prescribes for gfx950) is committed as profiles/<round>_bench_traffic.json and embedded
        traffic, traffic_src = None, None
        for tf in ("r05_bench_traffic.json", "r04_bench_traffic.json", "r03_bench_traffic.json", "r02_bench_traffic.json"):
            tf = os.path.join(ROOT, "profiles", tf)
            if traffic is None and os.path.exists(tf) and not decode:
                try:
                    rec = json.load(open(tf)).get("%s_%s" % (args.cell, args.dtype))
                    if rec and rec.get("T") == T and rec.get("B") == B:
                        # (measured per T-step problem under counter collection, where the layers run as single launches; a phase
                        #  launch holds lpl such problems)
                        traffic, traffic_src = rec["bytes_per_launch"] * lpl, rec
                except (ValueError, OSError):
                    pass
        fwd_flop = algorithmic_flops_per_window(spec)
        step_flop = (3.0 * fwd_flop if not decode else decoder_flops_per_window(spec)) * B
        bytes_per_row = (10 if args.cell == "LSTM" else 9) if not decode else (5 if args.cell == "LSTM" else 4)
        what = workload_name(args.config, C, seq, voices, T, latent, B, named_batch, named_gpus, args.cell, decode)
        out = {
            "metric": "MIDI roll windows/sec (%s)" % ("decode" if decode else "train step"), "value": B * world * args.steps / elapsed,
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "median_ms_per_step": median_ms, "step_ms": step_split, "arming_steps": ARM, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": what, "baseline_config": args.config, "global_batch": B * world, "T": T, "cell": args.cell,
                       "parallelism": ("replicas%d" if decode else "dp%d") % world},
            "roofline": {"bound": "mfma", "kernel": "%s (%s, %s %s, resident recurrent weights)" % (
                             dom, "forward recurrence" if decode else "BPTT", args.cell, args.dtype),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         # `traffic` per the contract is what THIS run measured: bench.py cannot run a counter pass over itself, so
                         # it is null; the committed PMC pass over the same kernel body is carried under its own name
                         "traffic": None, "traffic_from_committed_profile": traffic, "traffic_source": traffic_src,
                         # per time step and row: LSTM BPTT reads gates 4H + cell state H + upstream gradient H, writes da 4H;
                         # GRU reads gates 3H + h H + upstream gradient H, writes da 3H + r*h H; the
                         # inference forward reads x*W + b (G*H) and writes h (H)
                         "algorithmic_bytes_per_launch": lpl * Bp * T * H * bytes_per_row * (2 if args.dtype == "bf16" else 4),
                         "layers_per_launch": lpl,
                         "avg_launch_ms": avg_ms, "launches": launches,
                         "avg_steps_per_launch": tot_steps / launches, "us_per_time_step": us_dom,
                         "flop_per_time_step": flop_step,
                         # the recurrence is latency-bound by design: B/16 workgroups, one per CU, step after step
                         "cus_occupied": cus, "frac_of_occupied_cus": achieved / (peak * min(cus, 256) / 256.0),
                         "launch_ms_by_layer": {"%s:%s" % k: ms for k, (n, ms, _) in prof.items()},
                         # the two figures beside the per-kernel fraction
                         "whole_step": {"algorithmic_tflop": step_flop / 1e12, "tflops": step_flop / (ms_step * 1e-3) / 1e12,
                                        "frac_of_peak": step_flop / (ms_step * 1e-3) / 1e12 / peak}},
        }
        if not decode:
            out["roofline"]["critical_path"] = {
                "what": "the four serial recurrence phases of a step (encoder forward, decoder forward, decoder BPTT, encoder BPTT): "
                        "T x the measured time per time step of a stacked layer",
                "us_per_step_fwd": us_fwd, "us_per_step_bwd": us_dom, "bound_ms": 2.0 * T * (us_fwd + us_dom) * 1e-3,
                "frac_of_step": 2.0 * T * (us_fwd + us_dom) * 1e-3 / ms_step}
            out["elbo"] = {"loss_after_warmup": first_loss, "loss_final": m["loss"], "kl": m["kl"], "notes_loss": m["notes_loss"]}
            if elbo_gpu is not None:
                diff = max(abs(g[k] - c[k]) for g, c in zip(elbo_gpu, elbo_cpu) for k in g)
                out["elbo"].update({
                    "what": "ELBO (Keras total loss) and its parts after each of %d optimizer steps, fresh epsilon per step, on the "
                            "first %d windows of this run's inputs from the same initial parameters: this engine (%s, the timed "
                            "schedule) beside float64 torch-CPU arithmetic (oracle/torch_cpu.py, pinned to the NumPy oracle)"
                            % (args.elbo_steps, EB, args.dtype),
                    "windows": EB, "steps": args.elbo_steps, "elbo_gpu": [g["loss"] for g in elbo_gpu],
                    "elbo_cpu": [c["loss"] for c in elbo_cpu], "parts_gpu_final": elbo_gpu[-1], "parts_cpu_final": elbo_cpu[-1],
                    "max_abs_diff": diff, "within_1e-3": diff <= 1e-3})
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = out["value"] / cpu["value"] if cpu.get("value") else None
        elif decode and solo:
            out["cpu_baseline"] = {"value": None, "unit": "windows/s", "cores": 0, "kind": "port",
                                   "sample": "not timed for the decode configuration: oracle/torch_cpu.py covers the train step (the "
                                             "default --config 1 run carries the CPU baseline)"}
        out["plan"] = _plan_block(eng, host_ms_per_step=host_s / args.steps * 1e3,
                           what="step plans (include/midivae_hip.h): steps of the timed region enqueued by ONE mvae_plan_run call "
                                "('replayed'), the bracketed ones (every 4th: HIP events around the dominant launches, Engine._timed) included")
        if dp_stats is not None:
            out["dp"] = dp_stats
        if world == 1 and not args.no_other_configs and args.config == 1 and args.dtype == "bf16" and not (args.batch or args.seq_len
                                                                                                          or args.voices or args.latent):
            # The other workloads, AFTER the headline region (which is exactly what it was without them), each in a process of its
            # own started from here: an engine built in a process that has already driven another engine's hardware queues runs
            # up to 14 % slower (measured at the reference's shape: 1.45 / 1.55 / 1.67 / 1.46 ms as the 1st .. 4th engine of one
            # process under 16 hardware queues - profiles/r05_q_engines_per_process.txt); a user's process builds ONE model.
            import subprocess
            del eng
            torch.cuda.empty_cache()
            others = []
            for cfg, cell_o, k, wu in ((1, "GRU" if args.cell == "LSTM" else "LSTM", 20, 12), (2, args.cell, 10, 12),
                                       (4, args.cell, 20, 12), (0, "GRU", 60, 12)):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", args.dtype,
                                        "--side", "%d,%s,%d,%d" % (cfg, cell_o, k, wu)] + (["--step-times"] if args.step_times else []),
                                       capture_output=True, text=True, timeout=600)
                    if args.step_times:
                        sys.stderr.write(r.stderr[-2000:])
                    if r.returncode != 0:
                        raise RuntimeError("exit %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1:] or ""))
                    others.append(json.loads(r.stdout.strip().splitlines()[-1]))
                except Exception as e:          # (a side measurement must never cost the headline line)
                    others.append({"baseline_config": cfg, "cell": cell_o, "error": "%s: %s" % (type(e).__name__, e)})
            out["other_configs"] = others
            # the PRODUCT path: what a reference user calls, host conversion and upload included
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fit_e2e_bench.py"), "--json", "--songs", "2",
                                    "--cell", args.cell], capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    raise RuntimeError("exit %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1:] or ""))
                out["fit_e2e"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                out["fit_e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
