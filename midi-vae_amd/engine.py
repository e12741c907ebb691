"""Device engine of the MIDI-VAE train / inference step (one GPU, one process).

Owns every resident buffer in HBM (parameters + optimizer state, packed weight copies, saved activations, gradient
workspaces, input staging) and sequences the C-ABI kernels of libmidivae_hip.so for
    train_step   = what ``autoencoder.fit`` does per minibatch     (reference vae_training.py:804-809)
    eval_step    = ``autoencoder.evaluate`` per minibatch           (reference vae_training.py:300)
    encode       = ``encoder.predict``                              (reference vae_training.py:289,795)
    decode       = ``decoder.predict`` (+ fused argmax decode)      (reference vae_evaluation.py:2482-2483)
torch supplies device memory, streams / events and graph capture only; every arithmetic operation of the step is a
kernel of the in-tree HIP library.  There is no CPU fallback: constructing an Engine without the library or without
a GPU raises.

Graph structure and the semantics it follows are documented in DESIGN.md; reference citations are on the methods.
"""
from __future__ import annotations

from collections import OrderedDict

import ctypes as C
import numpy as np
import os

import torch

from . import hiplib as hl
from . import ops
from . import plan
from .layout import ModelSpec, ParamLayout, init_params

from .engine_io import ArrayStaging, Results
from .engine_optional import OptionalGraph
from .engine_phases import PhaseLaunches
from .engine_plan import PlannedSteps
from .engine_grads import ParamGradients
from .engine_steps import TrainSteps
from .engine_buffers import Buffers, _Rec, _Head, _Aux        # noqa: F401  (the records of the layer description)
from .slots import *        # noqa: F401,F403  (scalar slots S_*, N_SCALARS, X_EXT)
from .slots import N_SCALARS, X_EXT, X_GATHER2


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Engine(Buffers, ArrayStaging, OptionalGraph, PhaseLaunches, PlannedSteps, ParamGradients, TrainSteps, Results):
    INDEX_DENSE = False      # the one-hot bottom layer's table rows written out inside the phase launch (_index_as_dense): measured, off
    def __init__(self, spec: ModelSpec, max_batch: int, dtype: str = "bf16", device: str = "cuda:0", seed: int = 0,
                 training: bool = True, share: "Engine | None" = None):
        """``share``: another Engine of the same spec on the same device whose PARAMETERS (the flat f32 buffer itself) and HIP
        streams this one uses - the forward-only engine model.py keeps beside the training engine for chip-filling inference
        batches (encoder.predict / evaluate / decoder.predict).  Its derived weight copies are its own;
        which of the two engines last prepared them for the current parameters is tracked by a shared version counter."""
        hl.load()          # raises HipLibraryMissing - no fallback
        if not torch.cuda.is_available():
            raise RuntimeError("the MIDI-VAE engine needs an MI355X (torch.cuda.is_available() is False)")
        self.spec, self.device, self.training = spec, torch.device(device), training
        self.kind = {"bf16": hl.BF16, "f32": hl.F32}[dtype]
        self.dt = ops.torch_dtype(self.kind)
        self.cell = hl.CELL_CODE[spec.cell]
        # the resident-weights recurrent kernels (H=256, bf16, GRU/LSTM) stream TILE16 sequences; everything else is
        # row-major.  Batches are padded to a multiple of 16 rows (one workgroup = 16 rows) with zero-weight rows.
        self.tile16 = (spec.H == 256 and self.kind == hl.BF16 and spec.cell in ("GRU", "LSTM"))
        #  Round 6: GRU layers on the two-waves-per-SIMD kernels (rnn_w8.hip, seq_layout TILE16Q): alone 1.23-1.54 us per time step
        #  forward, 1.88-2.08 BPTT against 1.50-1.75 / 2.16-2.37 for the one-wave kernels.  MVAE_GRU_W8=0: the round-5 kernels.
        self.gru_w8 = os.environ.get("MVAE_GRU_W8", "1") != "0"
        self.lay = hl.TILE16 if self.tile16 else hl.ROWMAJOR          # what the GEMM epilogues write (xp, dX)
        self.maxB = (int(max_batch) + 15) // 16 * 16
        self.layout = self._make_layout()
        self._plan_init()      # step plans + the event ring behind _fork / _join (engine_plan.py)
        self.prof = None       # dict -> per-kernel HIP-event pairs are recorded on the launch stream (bench.py)
        self.prof_kinds = None # None = every timed launch, else a set of kinds ("rnn_fwd", "rnn_bwd")
        L, dev = self.layout, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        if share is not None:
            assert not training and share.layout.total == L.total and share.device == self.device
            self.params, self._pver = share.params, share._pver
        else:
            self.params, self._pver = torch.zeros(L.total, **f32), [0]
        self._prepared_ver = -1
        n_opt = L.total if training else 0              # a forward-only engine holds no gradients / optimizer state
        self.grads = torch.zeros(n_opt, **f32)
        self.opt_m = torch.zeros(n_opt, **f32)
        self.opt_v = torch.zeros(n_opt, **f32)
        self.t_done = torch.zeros(1, dtype=torch.int32, device=dev)
        self.P = {n: L.view(self.params, n) for n in L.entries}
        self.G = {n: L.view(self.grads, n) for n in L.entries} if training else {}
        self.scal = torch.zeros(N_SCALARS, **f32)
        # Each recurrent kernel occupies B/16 CUs of 256, so the independent branches of the graph (notes stack /
        # velocity / instrument, forward and backward) run on their own HIP streams, and parameter-gradient GEMMs
        # (needed only by the optimizer) go to two more streams off the critical path.  Fork / join is event based.
        # A forward-only engine beside a training engine uses THAT engine's streams: a second set would alias onto the same
        # hardware queues, and two streams of one pipelined stack on one queue is the one thing the schedule cannot take.
        nl = max(spec.Le, spec.Ld) - 1
        if share is not None:
            self.s_vel, self.s_instr, self.s_grad, self.s_grad2 = share.s_vel, share.s_instr, share.s_grad, share.s_grad2
            self.s_held, self.s_next = share.s_held, share.s_next
            self.s_layer, self.s_proj = share.s_layer, share.s_proj
        else:
            with torch.cuda.device(self.device):
                self.s_vel, self.s_instr, self.s_grad = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
                self.s_held = torch.cuda.Stream() if spec.meta_held else None
                self.s_next = torch.cuda.Stream() if spec.meta_next else None
                self.s_grad2 = torch.cuda.Stream()      # second gradient stream: input-kernel / bias gradients
                self.s_layer = [torch.cuda.Stream() for _ in range(nl)]
                self.s_proj = [self._own_queue_stream() for _ in range(nl)]     # x*W / dX of pipelined stacks
        self.multi_stream = True
        self._prefork = None
        self._branches_stay_forked = False
        self.lean_sync = True            # fork / join with one packet on the critical queue (settled: r01 9.3 -> 8.86 ms; an attribute for tests)
        self._bucket_hook = None         # data parallel: dp.BucketedAllReduce of the running train_step
        self.s_comm = None               # ... and the stream its early bucket starts on (created on first use)
        self.status_allreduce = None     # data parallel: MAX of the pipeline status word over the ranks (dp.DataParallel)
        # Stacked layers are pipelined over TIME CHUNKS: layer l runs chunk k (on its own stream) as soon as layer l-1
        # has produced it, instead of waiting for the whole sequence.  The f32 state is carried across launches.
        self.time_chunks = 4
        # ... and where the slot-interleaved LSTM / GRU kernels apply AND the stack's kernels fit on the chip together
        # (_pipelined), as ONE launch per layer with device-side hand-over every pipe_chunk time steps (no relaunch, no weight
        # reload, layers pipe_chunk steps apart instead of T/4)
        self.pipeline = os.environ.get("MVAE_PIPELINE", "1") == "1"     # (0: one launch per (layer, chunk), e.g. several processes on ONE GPU)
        if getattr(self, "_serial_queues", False) or (share is not None and not share.pipeline and getattr(share, "_serial_queues", False)):
            self.pipeline = False
        # time steps per hand-over: a hand-over costs every workgroup a drained vmcnt and a counter, the persistent GEMM a wait - the
        # bigger the batch, the more rows a chunk should carry (A/B r02: 256 windows 8 / 16 / 32 / 64 -> 8.18 / 8.08 / 8.13 / 8.31 ms;
        # 512 windows, T=2048: 35.7 / 34.5 / 35.4 at 16 / 32 / 64; decode of 1024 windows: 57.0 / 58.9 / 60.1 / 59.9 k at 16 / 32 / 64 / 128)
        self.pipe_chunk = 16 if self.maxB <= 256 else 32 if self.maxB <= 512 else 64
        while self.pipe_chunk > 16 and spec.T % self.pipe_chunk:
            self.pipe_chunk //= 2
        # the HOST is held at these points of a train step until the device has reached them (_pace: bit 2 in front of the backward
        # pass, 4 in front of the encoder BPTT; 1 in front of the decoder forward): a device whose queues already hold the packets of
        # the later phases - value waits on a dozen queues - runs the current phase slower.  T = 64: 1.71 -> 1.45 ms per step with
        # 6, configs[1] 6.65 -> 6.57 (profiles/r05_c_steps_in_flight.txt); a replayed step is split into call ranges there
        self.pace_mask = 6
        # ... of a step enqueued in TWO calls (train_step_begin / train_step_finish: model.Autoencoder.fit, whose host converts the
        # next minibatch - or the caller's next song - between two steps): only the pause in front of the backward pass.  A host that
        # is held until the encoder BPTT starts comes back too late for that work: `python vae_training.py` at default settings
        # 58 k -> 71 k windows/s end to end (no pause at all: 75 k, but configs[1] GRU fit 5.8 -> 7.6 ms per step); fit at configs[1]
        # LSTM 7.9-8.1 -> 7.2-7.7 ms per 256-window step, GRU 5.64 -> 5.80 (profiles/r05_t_pace_mask_by_caller.txt)
        self.pace_mask_split = 2
        self._pace_mask_now = self.pace_mask
        self.pace_early = True            # (engine_steps._pace_point: the pauses end when the device reaches an EARLIER point of the queue)
        self._pace_events, self._pace_recorded = {}, set()
        self.gate_pipe_gemms = False      # (engine_phases._launch_pipe_gemms: measured, off)
        self.pipe_gemm_blocks = 64       # persistent grid of the dX GEMM between two pipelined layers (backward)
        # ... and of the forward projection x*W + b: the weights-stationary kernel (csrc/gemm.hip proj_ws_k) wants a multiple of
        # 8 XCDs x (G*H / 128) column tiles - one workgroup per (XCD, column tile) keeps its weight panel in LDS for the whole launch:
        # 64 for LSTM, 48 for GRU.  (Round 1's kernel reloaded the panel per tile; decoder inference at 1024 windows was bound by it:
        # 8.2 us per decoder step at 64 workgroups, 6.2 at 128 - 4.1 now at 64)
        self.pipe_proj_blocks = 8 * max(spec.GH // 128, 1)
        # Parameter-gradient GEMMs once per layer (after its BPTT), NOT per time chunk: throughput GEMMs running beside the
        # latency-bound recurrences slow those down by more than the tail they would save.
        self._grads_clean = False
        self.fuse_head_bwd = True        # d(h sequence) of the output Denses from the head launch (mvae_head wc / dhs)
        self.fuse_bias_grad = True       # bias gradients from the recurrent-kernel gradient GEMM's pass over da (mvae_gemm colsum_b)
        self.fused_latent = True         # Dense chain around the latent as one launch each way (csrc/latent.hip)
        # Encoder stack (the LAST recurrence phase of a step): its weight-gradient GEMMs FOLLOW the running BPTT kernels chunk by chunk
        # (mvae_gemm k_wait: one resident workgroup per (output tile, K partition) accumulates in registers over all chunks) - what
        # is left when the recurrence ends is one chunk's share instead of four whole GEMMs (0.55 ms of the 0.77 ms tail).  All of
        # them are ONE launch (mvae_gemm_kstream_multi) on the second gradient queue - a queue each cost more than the tail saved -
        # and the other gradient work of that phase (velocity / instrument encoders) goes to the first one.
        # kstream_wgs workgroups per GEMM: they wait beside the recurrences, one per CU (_kstream_ok: residency).
        # SHORT sequences (the reference's shipped T = 64, settings.py:108-109): every weight-gradient GEMM is a 20-120 us launch
        # that is mostly fill and drain, a dozen of them beside and behind the recurrences were two thirds of what followed the last
        # BPTT, and their workgroups kept the recurrent launches (one workgroup per EMPTY CU) waiting for a place.  Up to
        # defer_grads_rows rows (T x padded batch) per sequence they are collected during the backward pass and leave as ONE launch
        # (mvae_gemm_multi) on the critical queue behind the last recurrence (_wgemm / _flush_deferred_gemms); above that the
        # GEMMs are long enough to be worth running beside the recurrences.  (profiles/r05_c_*; 0 = never)
        # Measured on T = 64 with the host paced (steps_in_flight / pace_mask below): 256 windows GRU 2.07 -> 1.47 ms per step, LSTM
        # 1.72 -> 1.69; 64 windows GRU 1.46 -> 1.18
        self.defer_grads_rows = int(os.environ.get("MVAE_DEFER_GRADS_ROWS", "32768"))
        # ... the decoder side's (and the latent block's) BESIDE the encoder BPTT launch, on the gradient queue, released by that
        # launch's first published chunk (_flush_deferred_gemms(early=...)): reference shape GRU 1.46 -> 1.39 ms, LSTM 1.705 -> 1.70;
        # from defer_early_rows rows per sequence (192 windows x 64 steps: 1.525 -> 1.46; 128 windows 1.324 -> 1.327; 64: 1.17 -> 1.18)
        self.defer_early, self.defer_early_rows = True, 12288
        # ... with K split further until a launch has about this many workgroups (_flush_deferred_gemms): at T*B = 16384 rows the
        # usual split (2) makes ~150 workgroups of 128 k tiles each.  0 / 256 / 384 / 512 / 1024: GRU 1.405 / 1.39 / 1.40 / 1.338 / 1.383
        # ms per step at the reference's shape, LSTM 1.70-1.71 throughout
        self.defer_split_wgs = 512
        self._deferred_gemms, self._deferred_small = None, []
        self.kstream_grads = os.environ.get("MVAE_KSTREAM_GRADS", "1") == "1"
        # (GRU: 3 GEMMs per layer -> 8 problems in the encoder launch.  Round 5, 256 windows T=512: 24 workgroups each = 192 resident
        #  beside the 64 of the recurrences left the decoder side's held-back GEMMs no CU - 5.66 ms per step; 20 / 16 / 14 / 12:
        #  5.42 / 5.43 / 5.36 / 5.34-5.39 (12-15 = one workgroup per output tile: under a tracer they fall 0.26 ms behind the BPTT);
        #  <= 10: a workgroup owns two tiles and streams the second one after the BPTT: 6.13.  LSTM: 16 / 24 / 32 all 6.53, <= 12: 7.27)
        self.kstream_wgs = 16 if spec.cell == "GRU" else 32
        self.kstream_max_B = 256
        self.flush_before_join = True     # (engine.backward: the encoder's collected GEMMs in front of the gradient queues' joins)
        # k rows per workgroup and chunk of a K-streaming GEMM (0: the round-2 rule, kstream_wgs workgroups per GEMM whatever its tile
        # count).  Round 6, same box: GRU 4.84 -> 4.69 ms per step, LSTM 6.29 -> 6.28; 1024 rows: GRU 4.94 (profiles/r06_k_kstream_rows.txt)
        self.kstream_rows = 2048
        self.dec_kstream = True           # the decoder notes stack's too (engine_phases._notes_backward_multi, _dec_kstream_ok)
        self.hold_side_heads, self._hold_side = True, False
        self.kstream_singles = True      # (settled r02: LSTM 7.60 -> 7.50 ms, GRU 6.55 -> 6.38)    # ... and the dU GEMM of a full-length single-layer encoder branch
        self._kstream_extra = None
        self._grad_streams = None        # (s_grad, s_grad2) unless overridden for a phase
        self._n_side = 0                 # recurrences running beside the stack being scheduled (residency, _pipelined)
        self._cur_B = self.maxB          # padded batch of the call being scheduled
        self.num_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        self._occ = {k: max(1, hl.load().mvae_occupancy(i)) for i, k in enumerate(("dx", "proj", "kstream"))}
        self._chain_refused = False      # the library refused the fused latent chain once: steps key their plans by the exact window count
        self._hist_fused = None          # train step whose history comes out of its own encoder forward (model.py: fused pre-pass)
        self._redo_hist = None           # ... kept until the step is verified (_redo_step)
        self._fused_dst = None           # (caller's rows, engine buffer) of a fused pre-pass whose z' is still to be handed over
        # the recurrences of a phase as ONE launch on the critical queue instead of one launch per queue (engine_phases.py)
        self.phase_multi = True
        # ... up to this many (padded) windows per call: measured (profiles/r03_j_*) 256 windows -6 % (LSTM) / -9 % (GRU) per train
        # step, but 512 windows at T=2048 +6.5 % and decoding 1024 windows +4 % - there the recurrences themselves fill the chip
        # and the per-queue launches (producers dispatched first) place them better than one launch's index order
        self.phase_max_B = 256
        # (_index_as_dense; measured r03_r: GRU step -0.07 ms, LSTM +0.09 ms - there the bottom layer does not set the pace)
        #  Round 4: the indexed kernels gather tile PAIRS from a paired-column table (8 x 16 bytes per row and step instead of 16 x 8:
        #  LSTM 2.66 -> 2.17, GRU 2.06 -> 1.70 us per step alone - faster than a dense input) and the written-out rows lose on both
        #  cells (GRU 5.71 vs 5.63 ms per step, LSTM 7.11 vs 6.91: profiles/r04_l_index_dense_ab.txt): off by default
        self.index_dense = type(self).INDEX_DENSE      # (a class attribute: the buffers it needs are allocated here)
        self.index_dense_blocks = 16
        self.xpand_blocks = 16
        self.gate_side_heads = True      # (settled r03_z: -0.03 ms)   # (decoder_forward: counter instead of event)
        self._last_stack_gate = None
        # (_head_forward: inference on per-queue pipelined stacks.  profiles/r03_zz_decode_head_slices.txt: decode configs[4] LSTM
        #  -3..5 % with 2 slices, -1..3 % with 4; GRU -3 % / -6..7 %: the slices take memory bandwidth from the recurrences they follow)
        self.head_slices = 2 if spec.cell == "LSTM" else 4
        self._top_publish = None
        # (_join; r03_z: -0.02 / -0.04 ms.  NOT when kernels are run one at a time - rocprofv3 counter collection: a critical queue parked
        #  in a value wait and a writer queue held back behind it never finish; event joins work there)
        serial = getattr(self, "_serial_queues", False) or (share is not None and getattr(share, "_serial_queues", False))
        self.value_join = not serial and len(self.s_proj) > 0     # (no stacked layer: no probe was run)
        self._join_seq = {}
        self._diag_no_param_grads = os.environ.get("MVAE_DIAG_NO_PARAM_GRADS", "0") == "1"
        # (_grad_portions) -1: by the rows of the sequence (4 portions from 2^20 rows, none below 2^19), 0: off, N: N portions
        self.grad_portions = -1
        self._grad_portion_jobs = None
        self._single_slot, self._single_slot_end = 0, 3
        self._hold_dec_grads = 1     # (engine_phases._notes_backward_multi; A/B r03_j: LSTM -0.06 ms, GRU neutral)
        self._after_chain = None
        self._tail_streams = []          # queues besides the two gradient queues that carry gradient work of the running step
        if share is None:
            self.set_params(self._initial_params(seed))
        self._build_graph_description()
        self._alloc(self.maxB)
        self._views_cache = {}
        self._prep = None                # PrepBatch per (step count pending): prepare_weights
        self._zero_scal_job = None
        self._zero_grads_job = None
        self._deferred_side = None       # (phase launches) side-queue work to be released by a device counter instead of an event
        self._xp0_bias = set()           # constant-input cells whose xp0 rows hold the bias (written by the last weight preparation)
        self.start_zero = {}             # layer prefix -> the staged start rows of its head are all zero (staging)
        self._count_only = None
        self._count_pending = False
        self._sync_cum = {}
        self._pipe_verified = set()
        self._pipe_used = False          # a time-pipelined stack has been launched since the engine was built
        self._dxp0_clean = False
        self.norm_B = float(self.maxB)   # windows the batch-mean losses are normalised by (the GLOBAL minibatch under data parallelism)
        self._have_staged_targets = False
        self.acc = torch.zeros(N_SCALARS, dtype=torch.float32, device=self.device)     # epoch accumulators (accumulate_metrics)

    def _own_queue_stream(self):
        """a new stream that does NOT share its hardware queue with the current (critical) stream: the runtime deals streams onto
        GPU_MAX_HW_QUEUES queues round-robin, and a phase launch on the critical stream holds both the producer and the consumer
        of the persistent GEMM running on this one (engine_phases.py) - on one queue they would wait for each other until the
        time-out.  Asked of the runtime by experiment (mvae_streams_alias), once, here."""
        cur = torch.cuda.current_stream()
        scratch = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._aliased = getattr(self, "_aliased", [])          # (kept alive: a released stream's queue slot would be dealt again)
        for attempt in range(1, 9):
            st = torch.cuda.Stream()
            rc = hl.load().mvae_streams_alias(cur.cuda_stream, st.cuda_stream, scratch.data_ptr(), attempt)
            if rc == 0:
                return st
            hl.check(min(rc, 0), "mvae_streams_alias")
            self._aliased.append(st)
        # Eight streams in a row that cannot run beside the critical one: kernels are being run ONE AT A TIME in this process
        # (rocprofv3 counter collection does that).  Kernels that wait for each other cannot work then - say so now instead of
        # through two 2-4 s time-outs: chunk-per-launch schedule.
        import warnings
        warnings.warn("kernels on different streams do not run concurrently in this process (counter collection?): time-pipelined "
                      "stacks and phase launches are off (Engine.pipeline = False)")
        self._serial_queues = True
        return self._aliased.pop()

    @property
    def _weights_dirty(self):
        """the derived weight copies of THIS engine are older than the parameters (shared version counter: an optimizer step or
        set_params on the training engine also invalidates the copies of the forward-only engine that shares its parameters)"""
        return self._prepared_ver != self._pver[0]

    @_weights_dirty.setter
    def _weights_dirty(self, dirty):
        if dirty:
            self._pver[0] += 1
        else:
            self._prepared_ver = self._pver[0]

    # (hooks of the style-classifier engine, classifier.py: the same recurrent / head / optimizer machinery on another graph)
    def _make_layout(self):
        return ParamLayout.build(self.spec)

    def _initial_params(self, seed):
        return init_params(self.spec, seed)

    def _seq_layout(self, r):
        """Sequence layout (= kernel family) of one recurrent layer: the slot-interleaved LSTM / GRU kernels (TILE16P
        saved activations) wherever they apply, else the phased resident kernels (TILE16), else the generic row-major
        ones."""
        if not self.tile16:
            return hl.ROWMAJOR
        if self.spec.cell == "GRU" and self.gru_w8:      # (round 6) two waves per SIMD: rnn_w8.hip
            return hl.TILE16Q
        if self.spec.cell in ("LSTM", "GRU"):   # (a 1-feature input is expanded to x*W + b first: _scalar_as_dense)
            return hl.TILE16P
        return hl.TILE16

    def _il(self, r):
        """does the layer run on the slot-interleaved kernel family (one wave per SIMD: TILE16P; two: TILE16Q)?"""
        return self._seq_layout(r) in (hl.TILE16P, hl.TILE16Q)

    def _rnn_waves(self, r):
        """waves per workgroup of the layer's recurrent kernels that publish a chunk of a time-pipelined stack"""
        return hl.load().mvae_rnn_producer_waves(self._seq_layout(r))

    def _paired_table(self, r):
        """the lookup table of a one-hot input layer in the column order the slot-interleaved LSTM / GRU kernels gather (two unit tiles
        per 16-byte access: 8 gathers per row and step instead of 16, include/midivae_hip.h mvae_rnn_fwd_args.table_layout)"""
        return r.xmode == hl.X_INDEX and self.spec.cell in ("LSTM", "GRU") and self._il(r)

    def _scalar_as_dense(self, r):
        """1-feature input layers (velocity roll) of an LSTM / GRU model: x*W + b is written out (T*B*G*H bf16, one streaming
        kernel, ~0.1 ms) so that the layer runs on the slot-interleaved dense-input kernels (2.1 instead of 3.9 us/step)."""
        return r.xmode == hl.X_SCALAR and self._il(r)

    def _index_as_dense(self, r):
        """the bottom layer of the encoder notes stack (one-hot pitch rows) sets the pace of the encoder-forward phase, and the
        indexed-input kernel - a dependent 2 KB table gather per row and step - takes 2.67 us per time step against 2.22 for
        a dense input: inside a phase launch the table rows are written out by a chunk-publishing producer (PhaseLaunches.
        _xpand_problem) and the layer reads them as a dense projection."""
        return (self.index_dense and self.training and r.xmode == hl.X_INDEX and self.enc_notes and r is self.enc_notes[0] and
                self._il(r))

    def _mark(self, name):
        """development: timestamp on the main stream at a section boundary (``self.marks = []`` to collect): (name, host time of
        the enqueue, event) - tools/host_vs_device.py"""
        if getattr(self, "marks", None) is not None:
            import time
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, time.perf_counter(), e))

    # ---- stream helpers -------------------------------------------------------------------------------------
    def _fork(self, *streams):
        if self._prefork is not None and self._prefork[0] == torch.cuda.current_stream() and \
                all(any(st is p for p in self._prefork[1]) for st in streams):
            for st in streams:              # already forked by _fork_with_stack
                self._prefork[1].remove(st)
            return
        cur = torch.cuda.current_stream()
        if self.lean_sync and len(streams) > 1:
            ev = self._ev_record(cur)       # ONE marker packet on this queue, however many streams branch off
            for st in streams:
                self._ev_wait(st, ev)
            return
        for st in streams:
            self._wait_stream(st, cur)

    def _fork_with_stack(self, layers, *streams, also=()):
        """fork ``streams`` AND the streams a pipelined stack over ``layers`` will use (and ``also``) with one event; the
        forks the callees then ask for from THIS stream for those streams are skipped, once each (nothing they depend on
        may be launched on the current stream in between; the caller clears ``_prefork`` after the stack)."""
        extra = ()
        if self.lean_sync and self._pipelined(layers):
            L = len(layers)
            extra = (*self.s_layer[:L - 1], *self.s_proj[:L - 1], *also)
        self._fork(*streams, *extra)
        self._prefork = (torch.cuda.current_stream(), list(extra)) if extra else None

    def _join(self, *streams, word=None):
        """the current stream waits for ``streams``.  ``word`` (an index into the engine's join words; MVAE_VALUE_JOIN=1): the last
        hop is a value written behind the side queue's work (hipStreamWriteValue32) and waited for here (hipStreamWaitValue32)
        instead of an event record + a barrier packet on that event"""
        cur = torch.cuda.current_stream()
        if word is not None and self.value_join and self.multi_stream and streams:
            for a, b in zip(streams, streams[1:]):
                self._wait_stream(b, a)
            self._join_seq[word] = seq = self._join_seq.get(word, 0) + 1
            if seq >= self._REBASE:
                torch.cuda.synchronize()
                self.store["join_words"][word:word + 1].zero_()
                self._join_seq[word] = seq = 1
            seq = ops.CounterValue(seq, ("join", word))
            w = self.store["join_words"][word:word + 1]
            ops.stream_write_value32(w, seq, stream=streams[-1])
            ops.stream_wait_value32(w, seq, stream=cur)
            return
        if self.lean_sync and len(streams) > 1:
            # chained: every side queue takes its barrier packet when ITS work ends; the joining queue (the critical one)
            # processes one barrier packet instead of len(streams)
            for a, b in zip(streams, streams[1:]):
                self._wait_stream(b, a)
            self._wait_stream(cur, streams[-1])
            return
        for st in streams:
            self._wait_stream(cur, st)

    def _join_into(self, stream):
        """``stream`` waits for everything enqueued so far on the current stream and on every side stream"""
        self._wait_stream(stream, torch.cuda.current_stream())
        for st in (*self._side_streams(), self.s_grad, self.s_grad2, *self.s_layer, *self.s_proj):
            self._wait_stream(stream, st)

    def _side_streams(self):
        """the streams of the independent encoder / decoder branches beside the notes stack"""
        return [st for st in (self.s_vel, self.s_instr, self.s_held, self.s_next) if st is not None]

    def _side(self, fn):
        """Run ``fn`` (parameter-gradient work nobody waits for before the optimizer) on the second gradient stream,
        ordered after everything enqueued so far on the current stream.  While a list is collecting (``_deferred_side``: the
        phase launch that follows releases the work by a device counter - an event record here is one more packet, ~30 us, on
        the critical queue) the work is only noted."""
        if not self.multi_stream:
            fn()
            return
        if self._deferred_side is not None:
            self._deferred_side.append(fn)
            return
        self._wait_stream(self.s_grad2, torch.cuda.current_stream())
        with torch.cuda.stream(self.s_grad2):
            fn()

    def _on(self, stream):
        """context: run on ``stream`` (or stay on the current one when multi-stream execution is off)"""
        return torch.cuda.stream(stream) if self.multi_stream else _NullCtx()

    def _timed(self, key, fn, steps=0):
        """Run ``fn`` (one kernel launch); when profiling, bracket it with HIP events on the launch stream.
        ``prof_kinds`` (a set of kinds like "rnn_bwd" and / or full keys like ("rnn_bwd", "dec.notes.1")) limits which
        launches are bracketed: every event pair costs launch
        slots, and bench.py times its step with the dominant kernel's launches bracketed only."""
        if self.prof is None or (self.prof_kinds is not None and key[0] not in self.prof_kinds and key not in self.prof_kinds):
            fn()
            return
        # The pair comes from a pool indexed by the bracket's position in the call (C-ABI events, recorded through entry points a
        # step plan holds): a bracketed step records the same launch list with the same handles every time - and REPLAYS.  A pair's
        # previous measurement is read before it is recorded again (engine_plan._prof_harvest).
        i = self._prof_i
        self._prof_i += 1
        while len(self._prof_pool) <= i:
            pair = []
            for _ in range(2):
                h = C.c_void_p()
                hl.check(hl.load().mvae_event_create_timed(C.byref(h)), "mvae_event_create_timed")
                pair.append(h.value)
            self._prof_pool.append(pair)
        self._prof_harvest(i)
        e0, e1 = self._prof_pool[i]
        st = torch.cuda.current_stream().cuda_stream
        hl.check(hl.load().mvae_event_record(e0, st), "mvae_event_record")
        fn()
        hl.check(hl.load().mvae_event_record(e1, st), "mvae_event_record")
        self._prof_pending[i] = (key, steps)
        self._prof_call.append((i, key, steps))

    # ------------------------------------------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------------------------------------------
    def set_params(self, named: dict):
        self.params.copy_(torch.from_numpy(self.layout.pack(named)).to(self.device))
        self._weights_dirty = True

    def get_params(self) -> "OrderedDict[str, np.ndarray]":
        return self.layout.unpack(self.params.cpu().numpy())

    def get_grads(self) -> "OrderedDict[str, np.ndarray]":
        return self.layout.unpack(self.grads.cpu().numpy())

    def reset_optimizer(self):
        self.opt_m.zero_()
        self.opt_v.zero_()
        self.t_done.zero_()
        self._count_pending = False

    def _v(self, name, *shape):
        """View of buffer ``name`` with the given shape (the storage is sized for max_batch)."""
        n = int(np.prod(shape))
        return self.store[name][:n].view(*shape)

    # ------------------------------------------------------------------------------------------------------
    # weight preparation: packed / transposed / converted copies the kernels consume (once per optimizer step)
    # ------------------------------------------------------------------------------------------------------
    def prepare_weights(self):
        """Derived copies of the parameters the kernels consume (MFMA-fragment packed recurrent kernels, bf16 / transposed
        input kernels, one-hot lookup tables): ~25 jobs in ONE launch (mvae_prepare_batch) - as separate kernels they cost
        0.3 ms per step, 0.8 ms with several steps queued.  (Measured and dropped in round 3: also writing what depends on the
        step's inputs here - x*W + b of the velocity roll, start*W + b of the constant-input decoder cells - lengthens the launch
        everything waits for by more than the launches it replaces: +0.1 ms per step, profiles/r03_c_ab_prep_inputs.txt.)"""
        s, P = self.spec, self.P
        key = bool(self._count_pending)
        pb = self._prep.get(key) if self._prep else None
        if pb is None:                  # built once per (count pending): every source / destination is a fixed view
            self._prep = self._prep or {}
            pb = self._prep[key] = ops.PrepBatch()
            # the optimizer's step count rides along after a step (not while the status word of that step is set: its update was
            # skipped too); either way the status word is moved to the latched word and cleared for the step that starts here
            pb.add_i32(self.t_done, int(key), guard=self.store["pipe_status"], latch=self.store["pipe_latched"])
            for r in self.all_rec:
                p = r.prefix
                pb.pack_recurrent(P[p + ".U"], self.store[p + ".u_pack"], 0)
                if r.xmode == hl.X_INDEX:
                    pb.make_table(P[p + ".W"], P[p + ".b"], self._v(p + ".table", r.K, s.GH))
                    if (p + ".table_p") in self.store:
                        pb.make_table(P[p + ".W"], P[p + ".b"], self._v(p + ".table_p", r.K, s.GH),
                                      paired=8 if self._seq_layout(r) == hl.TILE16Q else True)
                elif r.xmode == X_GATHER2:
                    d0 = r.K - s.attach
                    pb.make_table(P[p + ".W"][:d0], P[p + ".b"], self._v(p + ".table", d0, s.GH))
                    pb.convert(P[p + ".W"][d0:], self._v(p + ".table2", s.attach, s.GH))
                elif r.xmode == hl.X_DENSE:
                    pb.transpose_convert(P[p + ".W"], self._v(p + ".wt", s.GH, s.H))
                if (p + ".wt2") in self.store:
                    pb.transpose_convert(P[p + ".W"], self._v(p + ".wt2", s.GH, 2 * s.H))
                    if self.training:
                        pb.convert(P[p + ".W"], self._v(p + ".wc2", 2 * s.H, s.GH))
                if self.training:
                    pb.pack_recurrent(P[p + ".U"], self.store[p + ".ut_pack"], 1)
                    if r.xmode == hl.X_DENSE:
                        pb.convert(P[p + ".W"], self._v(p + ".wc", s.H, s.GH))
            for h in self.heads:
                pb.transpose_convert(P[h.out + ".W"], self._v(h.name + ".wt", h.NP, s.H), n_pad=h.NP)
            for r in self.all_rec:          # start*W + b of the cells on a constant input, for an all-zero start (= the bias row)
                if r.xmode == hl.X_CONST:
                    pb.broadcast_rows(P[r.prefix + ".b"], self._v(r.prefix + ".xp0", self.maxB, s.GH), self.maxB)
            pb.zero(self.scal)              # the step's loss / metric accumulators (else a fill launch of its own)
            if self.training:
                for h in self.heads:
                    pb.convert_pad(P[h.out + ".W"], self.store[h.name + ".wc"], h.NP)
                for r in self.all_rec:      # sum over time of da of the constant-input cells: accumulated into a buffer this
                    if r.xmode == hl.X_CONST:       # launch zeroes (a hipMemsetAsync per layer and step otherwise)
                        pb.zero(self.store[r.prefix + ".dxp0"])
                for wname, tname in (("dec.init.W", "lat.wt_init"), ("enc.zmean.W", "lat.wt_mu"), ("enc.zlogvar.W", "lat.wt_lv"),
                                     ("enc.extra.W", "lat.wt_extra"), ("enc.pack.W", "lat.wt_pack")):
                    if tname in self.store:
                        pb.transpose_convert(P[wname], self.store[tname])
        pb.run()
        self._count_pending = False
        self._weights_dirty = False
        self._xp0_bias = {r.prefix for r in self.all_rec if r.xmode == hl.X_CONST}      # (their xp0 rows hold the bias now)

    # ------------------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------------------
    def _nchunks(self, layers):
        T = layers[0].T
        n = self.time_chunks if (len(layers) > 1 and self.multi_stream) else 1
        while n > 1 and (T % n or (T // n) < 16):
            n -= 1
        return n

    def _rec_forward(self, r, B, k=0, nch=1, *, h0=None, c0=None, h0_ld=0, h_last=None, h_last_ld=0, idx=None, xs=None,
                     start=None, pipe=None, xp_external=False, build=False):
        """Time chunk k of nch of one recurrent layer (B = padded batch).  Chunk 0 starts from (h0, c0); later chunks
        from the f32 state the previous launch left in <layer>.sh/.sc; the last chunk also writes ``h_last``.
        ``build``: return the launch's arguments as a problem of a phase launch (engine_phases.py) instead of launching."""
        s, P, p = self.spec, self.P, r.prefix
        H, GH, T = s.H, s.GH, r.T
        Tc = T // nch
        t0 = k * Tc
        sh, sc = self._v(p + ".sh", B, H), self._v(p + ".sc", B, H)
        kw = {}
        if r.xmode == hl.X_INDEX and xp_external:       # (the table rows written out inside the phase launch: _index_as_dense)
            kw.update(xp=self._v(p + ".xp", T, B, GH)[t0:t0 + Tc])
        elif r.xmode == hl.X_INDEX:
            if self._paired_table(r):
                kw.update(idx=idx[t0:t0 + Tc], table=self._v(p + ".table_p", r.K, GH),
                          table_layout=hl.TABLE_PAIRED8 if self._seq_layout(r) == hl.TILE16Q else hl.TABLE_PAIRED)
            else:
                kw.update(idx=idx[t0:t0 + Tc], table=self._v(p + ".table", r.K, GH))
        elif r.xmode == X_GATHER2:       # table[pitch] + table2[instrument] written out, then the dense-input kernels
            xp = self._v(p + ".xp", T, B, GH)[t0:t0 + Tc]
            ops.gather2_tile16(idx[t0:t0 + Tc], self._v("in.xa_idx", T, B)[t0:t0 + Tc], self.store[p + ".table"],
                               self.store[p + ".table2"], xp, Tc * B, GH, layout=self.lay)
            kw.update(xp=xp)
        elif self._scalar_as_dense(r):
            xp = self._v(p + ".xp", T, B, GH)[t0:t0 + Tc]
            if not xp_external:      # (a phase launch expands the roll inside the launch: PhaseLaunches._xpand_problem)
                ops.outer_bias_tile16(xs[t0:t0 + Tc], P[p + ".W"].view(-1), P[p + ".b"], xp, Tc * B, GH)
            kw.update(xp=xp)
        elif r.xmode == hl.X_SCALAR:
            kw.update(xs=xs[t0:t0 + Tc], w_row=P[p + ".W"].view(-1), bias=P[p + ".b"])
        elif r.xmode == X_EXT:
            kw.update(xp=self._v(p + ".xp", T, B, GH)[t0:t0 + Tc])
        elif r.xmode == hl.X_CONST:
            xp0 = self._v(p + ".xp0", B, GH)
            if k == 0 and not (self.start_zero.get(p, False) and p in self._xp0_bias):
                # start W + b (Appendix A.6); an all-zero start - what the reference's packers always pass - finds the bias rows
                # the weight-preparation launch left in xp0 (every row the same: any batch's view of the buffer holds them)
                ops.gemm(start, P[p + ".W"], xp0, B, GH, r.K, bias=P[p + ".b"])
                self._xp0_bias.discard(p)
            kw.update(xp0=xp0)
        else:
            xp = self._v(p + ".xp", T, B, GH)[t0:t0 + Tc]
            if not xp_external:      # (pipelined stacks produce xp chunk by chunk on the projection stream: _rec_xp)
                self._rec_xp(r, B, k, nch)
            kw.update(xp=xp)
        if pipe:
            kw.update(pipe)
        lstm = s.cell == "LSTM"
        last = k == nch - 1
        if k > 0:
            h0, c0, h0_ld = sh, (sc if lstm else None), 0
        launch = lambda **bk: ops.rnn_fwd(
            self.cell, self.kind, Tc, B, H, self.store[p + ".u_pack"], h0=h0, c0=c0, h0_ld=h0_ld,
            hs=self._v(p + ".hs", T + 1, B, H)[t0:t0 + Tc + 1],
            cs=self._v(p + ".cs", T + 1, B, H)[t0:t0 + Tc + 1] if (lstm and self.training) else None,   # (backward only)
            acts=self._v(p + ".acts", T, B, GH)[t0:t0 + Tc] if self.training else None,
            h_last=(h_last if last else sh), h_last_ld=(h_last_ld if last else 0),
            c_last=(sc if (lstm and not last) else None), seq_layout=self._seq_layout(r), **kw, **bk)
        if build:
            return launch(build_only=True)
        self._timed(("rnn_fwd", p), launch, steps=Tc)

    def _rec_xp(self, r, B, k=0, nch=1, **chunked):
        """Input projection x*W + b of a stacked layer (x = the lower layer's h sequence), time chunk k of nch; or, with
        the chunk_* arguments of ops.gemm, the whole sequence as ONE persistent launch (time-pipelined stacks)."""
        s, P, p = self.spec, self.P, r.prefix
        H, GH, T = s.H, s.GH, r.T
        Tc = T // nch
        t0 = k * Tc
        lower_hs = self._v(r.lower.prefix + ".hs", T + 1, B, H)[1 + t0:1 + t0 + Tc]
        xp = self._v(p + ".xp", T, B, GH)[t0:t0 + Tc]
        ops.gemm(lower_hs, self._v(p + ".wt", GH, H), xp, Tc * B, GH, H, trans_b=True, bias=P[p + ".b"], c_layout=self.lay,
                 **chunked)

    # ---- time-pipelined stacks (slot-interleaved LSTM kernels) ---------------------------------------------------
    def _resident_cus(self, layers, B, backward=None, side=False):
        """CUs the kernels of a time-pipelined stack over ``layers`` occupy AT THE SAME TIME at a padded batch of B windows
        (``side``: plus the ``_n_side`` single-layer recurrences of the phase running beside it).  A recurrent workgroup (16
        windows) owns a whole CU (__launch_bounds__(256, 1): 512 registers per lane, up to 160 KiB of LDS); the persistent
        projection / dX GEMM between two layers holds ceil(grid / workgroups per CU) more (occupancy from the library:
        mvae_occupancy)."""
        per, L = B // 16, len(layers)
        cus = {True: -(-self.pipe_gemm_blocks // self._occ["dx"]), False: -(-self.pipe_proj_blocks // self._occ["proj"])}
        g = max(cus.values()) if backward is None else cus[bool(backward)]
        return L * per + (L - 1) * g + (self._n_side * per if side else 0)

    def _pipelined(self, layers):
        """ONE launch per layer for the whole sequence; layer l+1 follows layer l at a distance of ``pipe_chunk`` time
        steps, released chunk by chunk through device-side counters (include/midivae_hip.h, 'time-pipelined stacks')
        instead of one launch per (layer, chunk).  Only when every kernel of the stack can be RESIDENT together: the hardware
        dispatches the engine's queues in no particular order, and a consumer (upper layer, projection GEMM) that takes its CUs
        first and waits for a producer that no longer finds a free one is the time-out the schedule must not depend on luck to
        avoid.  The phase's other recurrences (velocity / instrument branches) wait for nobody: beside an oversubscribed chip
        they only delay the stack until they retire (decode at 1024 windows runs that way, by measurement the best schedule
        there).  Chip-filling inference batches (2048 windows: 2 x 128 + 64 workgroups) run the layers as chunk launches ordered
        by events instead - nothing waits on the device there."""
        if len(layers) < 2:
            return False
        T = layers[0].T
        return (self.pipeline and self.multi_stream and len(layers) - 1 <= len(self.s_layer) and T % self.pipe_chunk == 0 and
                len(layers) * 2 * (T // self.pipe_chunk) <= 1024 and
                all(self._il(r) for r in layers) and
                self._resident_cus(layers, self._cur_B) <= self.num_cus)

    def _sync_region(self, slot, n_if, nchp, nwaves, pwaves):
        """[n_if][2][nchp] 32-bit counters (hs / da chunks published, xp / dX chunks published) of pipelined stack number
        ``slot`` and the values they will have reached when this call's producers are done.  The counters are never
        zeroed between calls - a fill kernel in front of every stack is one more dependent launch on the critical path -
        they count up monotonically and the thresholds move with them."""
        reg = self.store["sync"][slot * 1024: slot * 1024 + n_if * 2 * nchp].view(n_if, 2, nchp)
        cum = self._sync_cum.setdefault(slot, [0, 0])
        if cum[0] + nwaves >= self._REBASE or cum[1] + pwaves >= self._REBASE:      # (once in millions of steps; the plans' threshold)
            torch.cuda.synchronize()
            reg.zero_()
            cum[0] = cum[1] = 0
        cum[0] += nwaves
        cum[1] += pwaves
        # (step plans, engine_plan.py: the fields these values land in are patches - they travel as ops.CounterValue)
        return reg, ops.CounterValue(cum[0], ("sync", slot, 0)), ops.CounterValue(cum[1], ("sync", slot, 1))

    def _stack_forward_pipe(self, layers, B, slot, *, states=None, h_last=None, h_last_ld=0, idx=None, start=None, xs=None,
                            publish_top=False):
        cs = self.pipe_chunk
        T = layers[0].T
        nchp, nwaves, pwaves = T // cs, self._rnn_waves(layers[0]) * (B // 16), 4 * self.pipe_proj_blocks
        L = len(layers)
        sync, hs_target, xp_target = self._sync_region(slot, L - 1, nchp, nwaves, pwaves)
        status = self.store["pipe_status"]
        self._pipe_used = True
        lower_streams = self.s_layer[:L - 1]              # layer l < L-1 on lower_streams[l]; the top layer on this stream
        gemm_streams = self.s_proj[:L - 1]
        self._fork(*lower_streams, *gemm_streams)
        for li, r in enumerate(layers):
            top = li == L - 1
            st = states(r) if states else {}
            pipe = dict(chunk_steps=cs, status=status)
            if li > 0:
                pipe.update(wait_ready=sync[li - 1, 1], wait_value=xp_target)
            if not top:
                pipe["signal_done"] = sync[li, 0]
            elif publish_top:
                # (inference: the head follows the top layer slice by slice - _head_forward; a slot of its own: these counters
                #  advance only in calls that publish)
                sync_t, target_t, _ = self._sync_region(10, 1, nchp, nwaves, 0)
                pipe["signal_done"] = sync_t[0, 0]
                self._top_publish = (sync_t[0, 0], target_t, cs)
            def run():
                self._rec_forward(r, B, 0, 1, idx=idx, start=start, xs=xs, h_last=h_last if top else None,
                                  h_last_ld=h_last_ld if top else 0, pipe=pipe, xp_external=li > 0, **st)
            if top:
                run()
            else:
                with torch.cuda.stream(lower_streams[li]):
                    run()
                with torch.cuda.stream(gemm_streams[li]):     # projection for the layer above: one persistent launch
                    self._rec_xp(layers[li + 1], B, 0, 1, max_blocks=self.pipe_proj_blocks, chunk_rows=cs * B,
                                 chunk_wait=sync[li, 0], chunk_wait_value=hs_target, chunk_done=sync[li, 1], chunk_status=status)
        self._join(*lower_streams, *gemm_streams)

    def _stack_forward(self, layers, B, *, states=None, h_last=None, h_last_ld=0, idx=None, start=None, slot=0, xs=None,
                       publish_top=False):
        """A stack of recurrent layers, pipelined over time chunks: layer l on stream l."""
        self._cur_B = B
        if self._pipelined(layers):
            return self._stack_forward_pipe(layers, B, slot, states=states, h_last=h_last, h_last_ld=h_last_ld, idx=idx,
                                            start=start, xs=xs, publish_top=publish_top)
        nch = self._nchunks(layers)
        streams = [None] + self.s_layer[:len(layers) - 1]
        done = [[None] * nch for _ in layers]
        if nch > 1:
            self._fork(*streams[1:])
        for k in range(nch):
            for li, r in enumerate(layers):
                top = li == len(layers) - 1
                st = states(r) if states else {}
                def run():
                    if li > 0 and nch > 1:
                        self._ev_wait(torch.cuda.current_stream(), done[li - 1][k])
                    self._rec_forward(r, B, k, nch, idx=idx, start=start, xs=xs, h_last=h_last if top else None,
                                      h_last_ld=h_last_ld if top else 0, **st)
                    if nch > 1:
                        done[li][k] = self._ev_record(torch.cuda.current_stream())
                if li > 0 and nch > 1:
                    with torch.cuda.stream(streams[li]):
                        run()
                else:
                    run()
        if nch > 1:
            self._join(*streams[1:])

    def encoder_forward(self, B, with_init=False):
        """reference vae_definition.py:443-516 (encoder) incl. the KL layer :15-37 and sampling :498-502.  ``with_init``:
        decoder_forward on the sampled z follows immediately - the fused latent chain also writes the decoder's initial
        states."""
        s, P = self.spec, self.P
        H, Z = s.H, s.Z
        Breal, B = B, self.pad16(B)
        cat = self._v("cat", B, self.ncat * H)
        ldc = self.ncat * H
        self._cur_B, self._n_side = B, len(self.enc_meta)
        if not self._encoder_forward_multi(B, cat, ldc):      # (one launch for all of them: engine_phases.py)
            self._fork_with_stack(self.enc_notes, *[st for _, st, _ in self.enc_meta])
            for k, (r, st, src) in enumerate(self.enc_meta, 1):
                with self._on(st):
                    inp = (dict(xs=self._v(src, r.T, B)) if r.xmode == hl.X_SCALAR else dict(idx=self._v(src, r.T, B)))
                    self._rec_forward(r, B, h_last=cat[:, k * H:(k + 1) * H], h_last_ld=ldc, **inp)
            if self.enc_bi:
                self._enc_bi_forward(B, cat[:, 0:H], ldc)
            else:
                self._stack_forward(self.enc_notes, B, idx=self._v("in.x_idx", s.T, B), h_last=cat[:, 0:H], h_last_ld=ldc, slot=0)
            self._prefork = None
            self._join(*[st for _, st, _ in self.enc_meta])
        self._n_side = 0
        self._mark("  encoder recurrences")
        self._S_done = False
        fused_hist = self._hist_fused            # (history of this minibatch from this very forward pass: train_step_begin)
        if not (self.fused_latent and self._chain_ok() and
                self._latent_chain_forward(Breal, B, with_init and fused_hist is None)):
            self._chain_refused_now()
            self._latent_forward_unfused(Breal, B)
        if fused_hist is not None:
            eps2, z_out = fused_hist
            zh = self._v("zh", B, s.zin)
            ops.history_from_latent(self._v("mu", B, Z), self._v("lv", B, Z), eps2, ops.ParamInt(Breal, ops.PARAM_B), B, Z, zh[:, Z:2 * Z],
                                    z_out=z_out)
        self._signature_forward(Breal, B)

    def _chain_ok(self):
        """shapes the fused latent chain (csrc/latent.hip) takes: a pack Dense whenever rolls are concatenated, 16-byte rows"""
        s = self.spec
        return (self.has_pack or self.ncat == 1) and s.zin % 4 == 0 and s.Z % 4 == 0

    def _chain_refused_now(self):
        """The step runs the separate latent launches (the library refused the fused chain: its workgroup would need more than
        160 KB of LDS - H = 512 with several heads - or the graph has a signature head).  They take the real window count and the
        loss normaliser as plain arguments and zero the padding rows with a torch operation only on a ragged batch, so such a step
        must never share a plan between window counts (_kind_B) - and a recording made under the padded key before this was known
        is dropped."""
        if not self._chain_refused:
            self._chain_refused = True
            rec = plan.active()
            if rec is not None and rec.tainted is None:
                rec.tainted = "separate latent launches (the fused chain was refused)"

    def _latent_chain_forward(self, Breal, B, with_init):
        """Encoder tail Denses, latent block and the decoder's initial-state Denses as ONE launch (csrc/latent.hip): six
        dependent small kernels otherwise, in a stretch of the step where nothing else can run.  False = shape not
        supported by the library (the separate launches follow)."""
        s, P = self.spec, self.P
        H, Z = s.H, s.Z
        tg = s.style and self._have_targets
        ok = ops.latent_chain_fwd(
            B, ops.ParamInt(Breal, ops.PARAM_B), H, Z, s.C if s.style else 0, self.ncat, s.zin, self.n_init * H, s.split, s.beta, s.prior_mean,
            s.prior_std, ops.ParamFloat(1.0 / self.norm_B, ops.PARAM_INV_BATCH), cat=self._v("cat", B, self.ncat * H),
            w_pack=P["enc.pack.W"] if self.has_pack else None, b_pack=P["enc.pack.b"] if self.has_pack else None,
            w_extra=P["enc.extra.W"] if s.extra_layer else None, b_extra=P["enc.extra.b"] if s.extra_layer else None,
            w_mu=P["enc.zmean.W"], b_mu=P["enc.zmean.b"], w_lv=P["enc.zlogvar.W"], b_lv=P["enc.zlogvar.b"],
            w_init=P["dec.init.W"] if with_init else None, b_init=P["dec.init.b"] if with_init else None,
            eps=self._v("in.eps", B, Z), style_target=self._v("in.c_idx", Breal) if tg else None,
            style_row_weight=self._v("in.rw_style", Breal) if tg else None,
            pack=self._v("pack", B, H) if self.has_pack else None, extra=self._v("extra", B, H) if s.extra_layer else None,
            mu=self._v("mu", B, Z), logvar=self._v("lv", B, Z), zh=self._v("zh", B, s.zin),
            style_probs=self._v("style_p", B, s.C) if s.style else None, scalars=self.scal[S_KL:S_KL + 3],
            S=self._v("S", B, self.n_init * H) if with_init else None)
        if ok:
            self._tail = (self._v("extra", B, H) if s.extra_layer else self._v("pack", B, H) if self.has_pack
                          else self._v("cat", B, H))
            self._S_done = with_init
        return ok

    def decoder_forward(self, B, want_probs=False):
        """reference vae_definition.py:519-645 (decoder heads) + losses :332-441 when targets are staged."""
        s, P = self.spec, self.P
        H, T, V = s.H, s.T, s.V
        Breal, B = B, self.pad16(B)
        zh = self._v("zh", B, s.zin)
        S = self._v("S", B, self.n_init * H)
        if not getattr(self, "_S_done", False):      # (else written by the fused latent chain of encoder_forward)
            ops.gemm(zh, P["dec.init.W"], S, B, self.n_init * H, s.zin, bias=P["dec.init.b"], act=hl.ACT_TANH)
        self._S_done = False
        ldS = self.n_init * H

        def states(r):
            k = r.init_block
            h0 = S[:, k * H:(k + 1) * H]
            c0 = S[:, (k + 1) * H:(k + 2) * H] if s.cell == "LSTM" else None
            return dict(h0=h0, c0=c0, h0_ld=ldS)

        self._mark("  decoder initial states")
        tg = self._have_targets
        side = [h for h in self.dec_heads if h.stream is not None]
        aux_src = {a.src for a in self.aux}
        self._cur_B, self._n_side = B, len(side)
        multi = len(self.dec_notes) > 1 and self._phase_ok(self.dec_notes, ())      # (the notes stack is ONE launch on this queue)
        if multi and side and self.gate_side_heads:
            # the side heads leave the critical queue WITHOUT an event (a record is a packet of ~50 us between the latent chain and the
            # decoder launch): their queues wait for the first chunk the notes stack publishes - it runs behind the latent chain
            self._last_stack_gate = None
            self._head_forward(self.head["notes"], B, Breal, states, tg, want_probs or "notes" in aux_src, slot=1)
            assert self._last_stack_gate is not None, "the notes stack did not run as a phase launch"
            word, value = self._last_stack_gate
            for h in side:
                ops.stream_wait_value32(word, value, stream=h.stream)
        else:
            if multi:
                if side:
                    self._fork(*[h.stream for h in side])
            else:
                self._fork_with_stack(self.dec_notes, *[h.stream for h in side])
        for h in side:
            with self._on(h.stream):
                self._head_forward(h, B, Breal, states, tg, want_probs or h.name in aux_src, slot=None)
        if not (multi and side and self.gate_side_heads):
            self._head_forward(self.head["notes"], B, Breal, states, tg, want_probs or "notes" in aux_src, slot=1)
        self._prefork = None
        self._n_side = 0
        if not self._branches_stay_forked or self.aux:
            self._join(*[h.stream for h in side])
        for a in self.aux:
            self._aux_forward(a, B, Breal, tg, want_probs)

    def _head_forward(self, h, B, Breal, states, tg, want_probs, slot):
        """cell stack + output Dense / activation / loss / accuracy / argmax of one decoder head (B = padded batch)"""
        s, P = self.spec, self.P
        H, n = s.H, h.name
        start = self._v("in.start_" + n, B, h.layers[0].K)
        self._top_publish = None
        if len(h.layers) > 1 and slot is not None:
            multi = self._notes_forward_multi(h, B, states, start)      # (both layers as one launch: engine_phases.py)
            if multi and self.training:
                self._pace_point(2)          # (the backward pass may be enqueued while the notes head runs)
            if not multi:
                # Inference on the per-queue pipelined schedule (a decode of 1024 windows x 4096 steps: the head's pass over 2 GB
                # of h is 1.06 ms behind a 10.3 ms stack): the top layer publishes its chunks too and the head runs slice by
                # slice on the (idle) gradient queue, each slice released by the chunk that completes it - only the last slice
                # is left when the recurrence ends
                nsl = self.head_slices
                sliced = (not self.training and not tg and not want_probs and nsl > 1 and h.T % nsl == 0 and
                          (h.T // nsl) % self.pipe_chunk == 0)
                self._stack_forward(h.layers, B, states=states, start=start, slot=slot, publish_top=sliced)
        else:
            for r in h.layers:           # (a stack off the critical stream - the next-notes head - runs layer after layer)
                self._rec_forward(r, B, start=start, **states(r))
        top = self._v(h.layers[-1].prefix + ".hs", h.T + 1, B, H)[1:]
        R = h.T * B
        tgt = {}
        if tg:
            tgt = (dict(target_val=self._v(h.target, R)) if h.kind == 1 else dict(target_idx=self._v(h.target, R)))
            tgt["row_weight"] = self._v("in.rw_" + n, R)
            if n == "notes" and s.attach:
                tgt["target_idx2"] = self._v("in.ya_idx", R)
        if self._top_publish is not None:
            word, target, cs = self._top_publish
            self._top_publish = None
            nsl = self.head_slices
            Ts = h.T // nsl
            am = self._v(n + ".argmax", R)
            for i in range(nsl):
                r0, r1 = i * Ts * B, (i + 1) * Ts * B
                last = i == nsl - 1
                def run():
                    ops.head(h.kind, self.kind, r1 - r0, H, h.N, top[i * Ts:(i + 1) * Ts], self._v(n + ".wt", h.NP, H), P[h.out + ".b"],
                             grad_scale=h.weight, probs=None, argmax=am[r0:r1], dlogits=None,
                             scalars=self.scal[h.slot:h.slot + 2], b_stride=B, b_valid=ops.ParamInt(Breal, ops.PARAM_B))
                if last:        # (behind the stack on this queue: complete by queue order)
                    run()
                else:
                    k = (i + 1) * Ts // cs - 1
                    ops.stream_wait_value32(word[k:k + 1], target, stream=self.s_grad)
                    with torch.cuda.stream(self.s_grad):
                        run()
            self._join(self.s_grad, word=3)
            return
        ops.head(h.kind, self.kind, R, H, h.N, top, self._v(n + ".wt", h.NP, H), P[h.out + ".b"], grad_scale=h.weight,
                 probs=self._v("out.%s_p" % n, R, h.N) if want_probs else None, argmax=self._v(n + ".argmax", R),
                 dlogits=self._v(n + ".dl", R, h.NP) if (self.training and tg) else None, **self._fused_head_bwd(n, tg),
                 scalars=self.scal[h.slot:h.slot + 2], b_stride=B, b_valid=ops.ParamInt(Breal, ops.PARAM_B), **tgt)

    # ------------------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------------------
    def _split_k(self, K):
        # weight-gradient GEMMs have a tiny output (H x G*H = 16 tiles of 128x128) and K = T*B: split K so that
        # tiles x splits ~ the CU count; more splits only add atomic traffic (212 vs 367 TFLOP/s at 64 vs 16)
        return int(min(16, max(1, K // 8192)))

    def _rec_bptt(self, r, B, k=0, nch=1, *, dhs_ext=None, dh_last=None, dh_last_ld=0, dh0=None, dc0=None, dh0_ld=0,
                  pipe=None, build=False):
        """BPTT over time chunk k (chunks run from the LAST to the first) + the gradient for the layer below."""
        s, p = self.spec, r.prefix
        H, GH, T = s.H, s.GH, r.T
        Tc = T // nch
        t0 = k * Tc
        lstm = s.cell == "LSTM"
        sh, sc = self._v(p + ".sh", B, H), self._v(p + ".sc", B, H)
        first, final = k == nch - 1, k == 0
        da = self._v(p + ".da", T, B, GH)[t0:t0 + Tc]
        launch = lambda **bk: ops.rnn_bwd(
            self.cell, self.kind, Tc, B, H, self.store[p + ".ut_pack"], self._v(p + ".hs", T + 1, B, H)[t0:t0 + Tc + 1],
            self._v(p + ".cs", T + 1, B, H)[t0:t0 + Tc + 1] if lstm else None, self._v(p + ".acts", T, B, GH)[t0:t0 + Tc],
            da, dhs_ext=None if dhs_ext is None else dhs_ext[t0:t0 + Tc],
            dh_last=(dh_last if first else sh), dh_last_ld=(dh_last_ld if first else 0),
            dc_last=(None if (first or not lstm) else sc),
            rh=self._v(p + ".rh", T, B, H)[t0:t0 + Tc] if s.cell == "GRU" else None,
            dh0=(dh0 if final else sh), dc0=((dc0 if final else sc) if lstm else None), dh0_ld=(dh0_ld if final else 0),
            seq_layout=self._seq_layout(r), **(pipe or {}), **bk)
        if build:
            return launch(build_only=True)
        self._timed(("rnn_bwd", p), launch, steps=Tc)

    def _rec_dx(self, r, B, k=0, nch=1, **chunked):
        """Gradient w.r.t. the input sequence of layer ``r`` (= what the layer below receives at its h_t), time chunk k.
        Runs on the LOWER layer's stream, right before that layer's BPTT of the chunk: the upper layer's next chunk
        launches without waiting for it."""
        s, p = self.spec, r.prefix
        H, GH, T = s.H, s.GH, r.T
        Tc = T // nch
        t0 = k * Tc
        da = self._v(p + ".da", T, B, GH)[t0:t0 + Tc]
        dx = self._v(p + ".dx", T, B, H)[t0:t0 + Tc]
        ops.gemm(da.view(Tc * B, GH), self._v(p + ".wc", H, GH), dx, Tc * B, H, GH, trans_b=True, c_layout=self.lay,
                 **chunked)

    def _stack_backward_pipe(self, layers, B, slot, *, dhs_ext=None, dh_last=None, dh_last_ld=0, dstates=None, idx=None,
                             xs=None, start=None):
        cs = self.pipe_chunk
        T = layers[0].T
        nchp, nwaves, pwaves = T // cs, self._rnn_waves(layers[0]) * (B // 16), 4 * self.pipe_gemm_blocks
        order = list(reversed(layers))               # order[0] = top layer: runs on this stream, publishes da
        L = len(order)
        kstream = self._kstream_ok(layers, B)
        if kstream:
            before = self._ev_record(torch.cuda.current_stream())      # the layers' saved inputs and the zeroed gradient buffers
        sync, da_target, dx_target = self._sync_region(slot, L, nchp, nwaves, pwaves)     # (row L-1: the bottom layer's da)
        status = self.store["pipe_status"]
        self._pipe_used = True
        lower_streams = self.s_layer[:L - 1]         # order[li], li >= 1, on lower_streams[li - 1]
        gemm_streams = self.s_proj[:L - 1]
        self._fork(*lower_streams, *gemm_streams)
        for li, r in enumerate(order):
            top = li == 0
            ds = dstates(r) if dstates else {}
            # EVERY layer publishes its da chunks, the bottom one too, whether or not a K-streaming launch follows it in THIS
            # call: the counters are cumulative over calls (_sync_region), so a row that is only advanced by some calls - the
            # K-streaming decision depends on the call's batch - would fall behind its thresholds for good (0.03 us per step)
            pipe = dict(chunk_steps=cs, status=status, signal_done=sync[li, 0])
            if li > 0:
                pipe.update(wait_ready=sync[li - 1, 1], wait_value=dx_target)
            def run():
                ext = dhs_ext if top else self._v(order[li - 1].prefix + ".dx", r.T, B, self.spec.H)
                self._rec_bptt(r, B, 0, 1, dhs_ext=ext, dh_last=dh_last if top else None,
                               dh_last_ld=dh_last_ld if top else 0, pipe=pipe, **ds)
                if not kstream:
                    self._rec_param_grads(r, B, idx=idx, xs=xs, start=start, publishes=(sync[li, 0], da_target, cs))
            if top:
                run()
            else:
                with torch.cuda.stream(lower_streams[li - 1]):
                    run()
            if li < L - 1:
                with torch.cuda.stream(gemm_streams[li]):     # dX for the layer below, from the last chunk to the first
                    self._rec_dx(r, B, 0, 1, max_blocks=self.pipe_gemm_blocks, chunk_rows=cs * B, chunk_reverse=True,
                                 chunk_wait=sync[li, 0], chunk_wait_value=da_target, chunk_done=sync[li, 1], chunk_status=status)
        # chained join: the latest finisher (the bottom layer) last - ahead of the dX GEMMs it would put two hops in series
        if kstream:     # behind every kernel of the stack in host order: the recurrences are dispatched first
            problems = []
            for li, r in enumerate(order):
                problems += self._kstream_problems(r, B, idx, dict(counters=sync[li, 0], target=da_target, rows=cs * B, status=status))
            gates = []
            for extra, word, value in (self._kstream_extra or ()):       # single-layer branches launched before this stack (backward)
                if len(problems) + len(extra) <= 8:
                    problems += extra
                    gates.append((word, value))
            self._kstream_extra = None
            self._ev_wait(self.s_grad2, before)
            for word, value in gates:
                ops.stream_wait_value32(word, value, stream=self.s_grad2)
            # Its workgroups wait, resident, for the whole BPTT: they may only take CUs once EVERY kernel they wait for is running
            # (a recurrent workgroup needs a whole empty CU) - the queue holds the launch back until the BOTTOM layer has
            # published its first chunk, which it can only do with the layers above it and the dX GEMMs running too.
            ops.stream_wait_value32(sync[L - 1, 0][nchp - 1:nchp], da_target, stream=self.s_grad2)
            with torch.cuda.stream(self.s_grad2):
                ops.gemm_kstream_multi(problems)
        self._join(*gemm_streams, *lower_streams)

    def _stack_backward(self, layers, B, *, dhs_ext=None, dh_last=None, dh_last_ld=0, dstates=None, idx=None, xs=None,
                        start=None, slot=0, kstream_extra=None):
        """BPTT through a stack (top layer first), pipelined over time chunks in reverse order."""
        self._cur_B = B
        if self._pipelined(layers):
            return self._stack_backward_pipe(layers, B, slot, dhs_ext=dhs_ext, dh_last=dh_last, dh_last_ld=dh_last_ld,
                                             dstates=dstates, idx=idx, xs=xs, start=start)
        nch = self._nchunks(layers)
        order = list(reversed(layers))               # order[0] = top layer
        if kstream_extra is not None and len(layers) == 1 and nch == 1:
            # a single-layer branch beside a K-streaming stack: ONE launch that publishes its da chunk by chunk (nobody waits inside
            # the stack), its dU GEMM joins the stack's K-streaming launch, the rest of its gradients follows its end as usual
            r, cs = layers[0], self.pipe_chunk
            sync, target, _ = self._sync_region(4, 1, r.T // cs, self._rnn_waves(r) * (B // 16), 0)
            self._rec_bptt(r, B, 0, 1, dhs_ext=dhs_ext, dh_last=dh_last, dh_last_ld=dh_last_ld,
                           pipe=dict(chunk_steps=cs, status=self.store["pipe_status"], signal_done=sync[0, 0]),
                           **(dstates(r) if dstates else {}))
            ks = dict(counters=sync[0, 0], target=target, rows=cs * B, status=self.store["pipe_status"])
            kstream_extra.append((self._kstream_problems(r, B, idx, ks, only_dU=True), sync[0, 0][r.T // cs - 1:r.T // cs], target))
            self._rec_param_grads(r, B, idx=idx, xs=xs, start=start, skip_dU=True)
            return
        if (len(layers) == 1 and nch == 1 and self._grad_portion_jobs is not None and self.pipeline and
                self._il(layers[0]) and layers[0].T % self.pipe_chunk == 0 and
                self._portion_count(layers[0], B, self.pipe_chunk) > 1 and self._single_slot < self._single_slot_end):
            # a full-length single-layer branch whose gradients go in time portions: ONE launch that publishes its da chunks
            r, cs = layers[0], self.pipe_chunk
            sync, target, _ = self._sync_region(11 + self._single_slot, 1, r.T // cs, self._rnn_waves(r) * (B // 16), 0)
            self._single_slot += 1
            self._pipe_used = True
            self._rec_bptt(r, B, 0, 1, dhs_ext=dhs_ext, dh_last=dh_last, dh_last_ld=dh_last_ld,
                           pipe=dict(chunk_steps=cs, status=self.store["pipe_status"], signal_done=sync[0, 0]),
                           **(dstates(r) if dstates else {}))
            self._rec_param_grads(r, B, idx=idx, xs=xs, start=start, publishes=(sync[0, 0], target, cs))
            return
        streams = [None] + self.s_layer[:len(layers) - 1]
        done = [[None] * nch for _ in order]
        if nch > 1:
            self._fork(*streams[1:])
        for k in range(nch - 1, -1, -1):
            for li, r in enumerate(order):
                top = li == 0
                ds = dstates(r) if dstates else {}
                def run():
                    if li > 0:
                        if nch > 1:
                            self._ev_wait(torch.cuda.current_stream(), done[li - 1][k])
                        self._rec_dx(order[li - 1], B, k, nch)
                    ext = dhs_ext if top else self._v(order[li - 1].prefix + ".dx", r.T, B, self.spec.H)
                    self._rec_bptt(r, B, k, nch, dhs_ext=ext, dh_last=dh_last if top else None,
                                   dh_last_ld=dh_last_ld if top else 0, **ds)
                    if nch > 1:
                        done[li][k] = self._ev_record(torch.cuda.current_stream())
                    if k == 0:
                        if self._hold_side and self._after_chain is not None and idx is None and xs is None:
                            # (a decoder side head beside a K-streaming notes stack: its gradient GEMMs would start at the end of
                            #  the phase and run across the latent chain - they wait for the encoder launch's first chunk instead)
                            self._after_chain.append(lambda r=r: dict(r=r, B=B, start=start))
                        else:
                            self._rec_param_grads(r, B, idx=idx, xs=xs, start=start)
                if li > 0 and nch > 1:
                    with torch.cuda.stream(streams[li]):
                        run()
                else:
                    run()
        if nch > 1:
            self._join(*streams[1:])

    def _fused_head_bwd(self, name, tg):
        """extra arguments of ops.head: the gradient w.r.t. the top cell's h sequence comes out of the head launch itself"""
        if not (self.training and tg and self.fuse_head_bwd and self.lay == hl.TILE16 and self.spec.H <= 256):
            return {}
        if any(a.src == name or a.key == name for a in self.aux):     # (d(logits) of that head changes after its launch)
            return {}
        return dict(wc=self.store[name + ".wc"], dhs=self.store[name + ".dhs"])

    def _head_stack_backward(self, h, B, dstates, slot):
        """output Dense backward + BPTT through one decoder head's cell stack"""
        start = self._v("in.start_" + h.name, B, h.layers[0].K)
        if len(h.layers) > 1 and slot is not None and self._phase_ok(h.layers, ()):
            dext, head_grads = self._head_backward(B, h.name, h.layers[-1], h.N, h.NP, h.out + ".W", h.out + ".b", defer=True)
            self._notes_backward_multi(h, B, dext, dstates, start, head_grads)
            return
        dext = self._head_backward(B, h.name, h.layers[-1], h.N, h.NP, h.out + ".W", h.out + ".b")
        if len(h.layers) > 1 and slot is not None:
            self._stack_backward(h.layers, B, dhs_ext=dext, start=start, dstates=dstates, slot=slot)
            return
        for r in reversed(h.layers):
            self._stack_backward([r], B, dhs_ext=dext, start=start, dstates=dstates)
            if r.lower is not None:
                self._rec_dx(r, B)
                dext = self._v(r.prefix + ".dx", r.T, B, self.spec.H)

    def _head_backward(self, B, name, r, N, NP, outW, outb, defer=False):
        """d(logits) -> gradient of the output Dense and of the top cell's h sequence.  ``defer``: return (dhs, the Dense's
        parameter-gradient work) instead of enqueueing the latter behind an event."""
        s, G = self.spec, self.G
        H, T = s.H, r.T
        R = T * B
        dl = self._v(name + ".dl", R, NP)
        top = self._v(r.prefix + ".hs", T + 1, B, H)[1:].reshape(R, H)
        dhs = self._v(name + ".dhs", T, B, H)
        if not self._fused_head_bwd(name, True):
            ops.gemm(dl, self._v(name + ".wt", NP, H), dhs, R, H, NP, c_layout=self.lay)   # dl (R,NP) W^T (NP,H); pad rows zero
        def grads():
            self._wgemm(top, dl, G[outW], H, N, R, ldb=NP, split_k=self._split_k(R))
            self._small(lambda: ops.colsum(dl, R, N, G[outb], ldx=NP))
        if defer:
            return dhs, grads
        if self._deferred_gemms is not None:        # (collected, not enqueued: no fork)
            grads()
            return dhs
        self._fork(self.s_grad)
        with self._on(self.s_grad):
            grads()
        return dhs

    def backward(self, B):
        """Reverse of encoder_forward/decoder_forward; accumulates into self.grads (zeroed by the caller)."""
        s, P, G = self.spec, self.P, self.G
        H, Z, T, V = s.H, s.Z, s.T, s.V
        Breal, B = B, self.pad16(B)
        dS = self._v("dS", B, self.n_init * H)
        ldS = self.n_init * H

        def dstates(r):
            k = r.init_block
            return dict(dh0=dS[:, k * H:(k + 1) * H], dc0=dS[:, (k + 1) * H:(k + 2) * H] if s.cell == "LSTM" else None,
                        dh0_ld=ldS)

        # ---- decoder: the notes stack and the side heads, independent branches -----------------------------------
        # (one event for the branches, the notes head's gradient GEMM and the notes stack's lower layers)
        side = [h for h in self.dec_heads if h.stream is not None]
        self._cur_B, self._n_side = B, len(side)
        self._deferred_gemms = [] if self._defers_grads(B) else None
        for a in self.aux:
            self._aux_backward(a, B)
        notes_multi = len(self.dec_notes) > 1 and self._phase_ok(self.dec_notes, ())     # (one launch, gradient work by counters)
        if not notes_multi and self.grad_portions:       # (per-queue schedule: long sequences release their gradient work in portions)
            self._grad_portion_jobs, self._single_slot, self._single_slot_end = [], 0, 3      # (decoder: sync slots 11..13)
        # (data parallel with the early bucket - _bucket_hook - : every decoder-side gradient has to be QUEUED when the bucket's
        #  collective is issued, in front of the encoder launch: nothing is held back behind it then -
        #  test_nothing_is_added_to_the_decoder_bucket_after_its_collective_was_issued)
        early_bucket = self._bucket_hook is not None and self.multi_stream and self.layout.dec_begin > 0
        self._after_chain = [] if (notes_multi and not self.enc_bi and len(self.enc_notes) > 1 and not early_bucket and
                                   self._phase_ok(self.enc_notes, [r for r, _, _ in self.enc_meta])) else None
        if self._branches_stay_forked:      # (train step: the side heads' queues go straight on with their own backward)
            if not notes_multi:
                self._fork_with_stack(self.dec_notes, also=(self.s_grad,))
        elif notes_multi:
            if side:
                self._fork(*[h.stream for h in side])
        else:
            self._fork_with_stack(self.dec_notes, *[h.stream for h in side], also=(self.s_grad,))
        if notes_multi and self._dec_kstream_ok(self.dec_notes, B):
            self._grad_streams = (self.s_grad, self.s_grad)      # the second gradient queue holds the notes stack's K-streaming launch
            self._hold_side = self.hold_side_heads
        for h in side:
            with self._on(h.stream):
                self._head_stack_backward(h, B, dstates, slot=None)
        self._hold_side = False
        self._head_stack_backward(self.head["notes"], B, dstates, slot=2)
        if notes_multi:
            self._pace_point(4)              # (the encoder phase may be enqueued while the latent chain runs)
        self._grad_streams = None
        self._prefork = None
        self._flush_grad_portions()
        self._join(*[h.stream for h in side], word=0)
        self._mark("  decoder BPTT")
        # (the signature head adds to d(z) between the initial-state Denses and the latent block: separate launches then)
        enc_multi = not self.enc_bi and self._phase_ok(self.enc_notes, [r for r, _, _ in self.enc_meta]) and len(self.enc_notes) > 1
        self._deferred_side = [] if (enc_multi and not early_bucket) else None       # (released by the encoder launch's first published chunk)
        dcat = (self._latent_chain_backward(Breal, B)
                if (self.fused_latent and self._chain_ok() and not s.signature) else None)
        if dcat is None:
            self._chain_refused_now()
            dcat = self._latent_backward_unfused(Breal, B)
        latent_grads, self._deferred_side = self._deferred_side, None
        self._pace(4)
        ldc = self.ncat * H
        self._mark("  latent block backward")
        hook = self._bucket_hook
        if hook is not None and self.multi_stream and self.layout.dec_begin > 0:
            # data parallel: every decoder-side gradient [dec_begin, total) is queued by now - start its all-reduce on the
            # communication stream, beside the encoder BPTT (dp.BucketedAllReduce)
            if self.s_comm is None:
                with torch.cuda.device(self.device):
                    self.s_comm = torch.cuda.Stream()
            self._join_into(self.s_comm)
            with torch.cuda.stream(self.s_comm):
                self._host_call("early", lambda: hook.early(self.grads[self.layout.dec_begin:self.layout.total]))
        # ---- encoder recurrences: the notes stack and the meta rolls, independent branches ---------------------------
        self._cur_B, self._n_side = B, len(self.enc_meta)
        enc_one_launch = bool(enc_multi and self._encoder_backward_multi(B, dcat, ldc, latent_grads))      # (one launch: engine_phases.py)
        if not enc_one_launch:
            assert not latent_grads
            if self.grad_portions:
                self._grad_portion_jobs, self._single_slot, self._single_slot_end = [], 3, 5      # (encoder: 14..15 - disjoint: the counters are cumulative per slot)
            ks_extra = None
            if not self.enc_bi and self._kstream_ok(self.enc_notes, B):
                self._grad_streams = (self.s_grad, self.s_grad)     # the second gradient queue holds the notes stack's K-streaming launch
                ks_extra = self._kstream_extra = []
            self._fork_with_stack(self.enc_notes, *[st for _, st, _ in self.enc_meta])
            for k, (r, st, src) in enumerate(self.enc_meta, 1):
                with self._on(st):
                    inp = (dict(xs=self._v(src, r.T, B)) if r.xmode == hl.X_SCALAR else dict(idx=self._v(src, r.T, B)))
                    follow = (ks_extra is not None and not ks_extra and self.kstream_singles and r.T == T and
                              self._il(r) and r.xmode != hl.X_CONST and
                              (2 if s.cell == "GRU" else 1) + (3 if s.cell == "GRU" else 2) * len(self.enc_notes) <= 8)
                    self._stack_backward([r], B, dh_last=dcat[:, k * H:(k + 1) * H], dh_last_ld=ldc,
                                         kstream_extra=ks_extra if follow else None, **inp)
            if self.enc_bi:
                self._enc_bi_backward(B, dcat[:, 0:H], ldc)
            else:
                self._stack_backward(self.enc_notes, B, dh_last=dcat[:, 0:H], dh_last_ld=ldc, idx=self._v("in.x_idx", T, B), slot=3)
            self._prefork = None
            self._grad_streams = None
            self._flush_grad_portions()
        self._n_side = 0
        # the two gradient queues finish last and together: chained, they would put two cross-queue hops in series - the
        # early finishers are chained into one of them, the other is waited for directly
        # Phase launches: the persistent projection / dX GEMMs on their own queue are NOT waited for here - the launch that consumed
        # their last chunk has ended on this queue, so they have written everything (they exit behind their last publish), and their
        # queue orders them against the next step's GEMMs; the encoder rolls' queues carried nothing in this phase.  Every stream
        # in this join is one more cross-queue hop (40-60 us each, in series) in front of the optimizer.
        tail = self._tail_streams if enc_multi else [st for _, st, _ in self.enc_meta]
        self._tail_streams = []
        # Round 6: a step that collects its weight-gradient GEMMs (defer_grads_rows) and ran every encoder recurrence as ONE launch on
        # this queue has everything its collected GEMMs read right here - they go out BEFORE the joins, beside what the gradient
        # queues still have to do (the decoder side's early launch and its small reductions), instead of behind two cross-queue hops
        # (reference shape: the critical queue sat 0.16 ms in those waits with 0.09 ms of its own work still to come).
        flush_first = self.flush_before_join and enc_one_launch and self._deferred_gemms is not None
        if flush_first:
            self._flush_deferred_gemms()
        self._join(*tail, self.s_grad, word=1)
        self._join(self.s_grad2, word=2)
        if not flush_first:
            self._flush_deferred_gemms()

    def _defers_grads(self, B):
        """does a step of B (padded) windows collect its weight-gradient GEMMs for one launch behind the last recurrence?"""
        return bool(self.training and self.tile16 and self.multi_stream and 0 < self.spec.T * B <= self.defer_grads_rows)

    def _latent_chain_backward(self, Breal, B):
        """The same as ONE launch (csrc/latent.hip) followed by the parameter-gradient GEMMs on the side streams; None if
        the library does not support the shape."""
        s, P, G = self.spec, self.P, self.G
        H, Z = s.H, s.Z
        ldS, ldc = self.n_init * H, self.ncat * H
        dS, S, zh = self._v("dS", B, ldS), self._v("S", B, ldS), self._v("zh", B, s.zin)
        dmu, dlv = self._v("dmu", B, Z), self._v("dlv", B, Z)
        d_extra, d_pack = self._v("dtail", B, H), self._v("dtail2", B, H)
        dcat = self._v("dcat", B, ldc)
        pk, ex, cat = self._v("pack", B, H), self._v("extra", B, H), self._v("cat", B, ldc)
        ok = ops.latent_chain_bwd(
            B, ops.ParamInt(Breal, ops.PARAM_B), H, Z, s.C if s.style else 0, self.ncat, s.zin, ldS, s.split, s.beta, s.prior_mean, s.prior_std, s.w_style,
            ops.ParamFloat(1.0 / self.norm_B, ops.PARAM_INV_BATCH), wt_pack=self.store.get("lat.wt_pack"), wt_extra=self.store.get("lat.wt_extra"),
            wt_mu=self.store["lat.wt_mu"], wt_lv=self.store["lat.wt_lv"], wt_init=self.store["lat.wt_init"], S=S,
            pack=pk if self.has_pack else None,
            extra=ex if s.extra_layer else None, mu=self._v("mu", B, Z), logvar=self._v("lv", B, Z), eps=self._v("in.eps", B, Z),
            style_probs=self._v("style_p", B, s.C) if s.style else None,
            style_target=self._v("in.c_idx", Breal) if s.style else None,
            style_row_weight=self._v("in.rw_style", Breal) if s.style else None, dS=dS, dzh=self._v("dzh", B, s.zin), dmu=dmu,
            dlogvar=dlv, d_extra=d_extra if s.extra_layer else None, d_pack=d_pack if self.has_pack else None, dcat=dcat)
        if not ok:
            return None
        h = self._tail
        h1w = H // 2 if s.split else H
        h2w = H - h1w if s.split else H

        def param_grads():
            ops.gemm(zh, dS, G["dec.init.W"], s.zin, ldS, B, trans_a=True, accumulate=True)
            ops.colsum(dS, B, ldS, G["dec.init.b"])
            ops.gemm(h, dmu, G["enc.zmean.W"], h1w, Z, B, trans_a=True, lda=H, accumulate=True)
            ops.colsum(dmu, B, Z, G["enc.zmean.b"])
            ops.gemm(h[:, h1w:] if s.split else h, dlv, G["enc.zlogvar.W"], h2w, Z, B, trans_a=True, lda=H, accumulate=True)
            ops.colsum(dlv, B, Z, G["enc.zlogvar.b"])
            if s.extra_layer:
                ops.gemm(pk if self.has_pack else cat, d_extra, G["enc.extra.W"], H, H, B, trans_a=True, accumulate=True)
                ops.colsum(d_extra, B, H, G["enc.extra.b"])
            if self.has_pack:
                ops.gemm(cat, d_pack, G["enc.pack.W"], ldc, H, B, trans_a=True, accumulate=True)
                ops.colsum(d_pack, B, H, G["enc.pack.b"])

        self._side(param_grads)
        return dcat

    # ------------------------------------------------------------------------------------------------------
    # steps
    # ------------------------------------------------------------------------------------------------------
    def _flush_count(self):
        """apply a step-count increment still owed to ``t_done`` (see optimizer_step)"""
        if self._count_pending:
            if self._count_only is None:
                self._count_only = ops.PrepBatch()
                self._count_only.add_i32(self.t_done, guard=self._guard())
            self._count_only.run()
            self._count_pending = False

    def _guard(self):
        """The status word of the time-pipelined stacks as the optimizer's guard: while it is non-zero (a kernel gave up waiting
        for its producer: that step's gradients are invalid) the update and the step count are skipped ON THE DEVICE, so the
        parameters and moments stay valid.  The word lives for ONE step: the weight-preparation launch of the next step moves it
        into ``pipe_latched`` (what check_pipeline / the metric reads report) and clears it - one time-out costs one update, not
        the rest of the epoch.  Under data parallelism the word is MAX-reduced over the ranks behind the gradient all-reduce
        (status_allreduce), so every rank skips or applies the same updates."""
        return self.store["pipe_status"] if self.pipeline else None

    def get_optimizer_state(self):
        return dict(m=self.opt_m.clone(), v=self.opt_v.clone(), t=self.t_done.clone(), pending=self._count_pending)

    def set_optimizer_state(self, st):
        """moments / step count of another Engine of the same spec (the flat layout is identical whatever the batch size)"""
        self.opt_m.copy_(st["m"])
        self.opt_v.copy_(st["v"])
        self.t_done.copy_(st["t"])
        self._count_pending = bool(st["pending"])

    def optimizer_step(self, grad_scale=1.0):
        s = self.spec
        # (data parallel: whenever the hook is set, whatever THIS rank's schedule - a rank that fell back to chunked launches
        #  must still take part in the collective the others issue)
        if self.status_allreduce is not None:
            self._host_call("status", lambda: self.status_allreduce(self.store["pipe_status"]))
        # (the gradients are zeroed as they are consumed: the next step starts without a 17 MB fill launch in front of it)
        if s.optimizer == "Adam":
            # the step count is bumped by the weight-preparation launch that follows anyway (one dependent launch less
            # between two steps)
            if self._count_pending:
                self._flush_count()
            ops.adam_step_dev(self.params, self.grads, self.opt_m, self.opt_v, s.lr, self.t_done, grad_scale=grad_scale,
                              zero_grad=True, keep_count=True, guard=self._guard())
            self._count_pending = True
        else:
            ops.rmsprop_step(self.params, self.grads, self.opt_v, s.lr, grad_scale=grad_scale, zero_grad=True, guard=self._guard())
        self._grads_clean = True
        self._weights_dirty = True

    def stager(self):
        if self._stager is None:
            from .staging import Stager
            self._stager = Stager(self)
        return self._stager


    def eval_step(self, B, want_probs=False):
        """Forward + losses only (``autoencoder.evaluate`` / ``autoencoder.predict``)."""
        self._planned(("eval", B, float(self.norm_B), bool(want_probs)), lambda: self._eval_step(B, want_probs), params=self._call_params(B))

    def _eval_step(self, B, want_probs):
        self._have_targets = True
        if self._weights_dirty:
            self.prepare_weights()
        else:
            self._zero_scal()
        self.encoder_forward(B, with_init=True)
        self.decoder_forward(B, want_probs=want_probs)
        self._verify_pipeline(lambda: (self.scal.zero_(), self.encoder_forward(B, with_init=True),
                                       self.decoder_forward(B, want_probs=want_probs)), key="predict")

    def encode(self, B):
        """``encoder.predict``: z (B,Z) device view (left block of [z|history])."""
        self._planned(("encode", B, float(self.norm_B)), lambda: self._encode(B), params=self._call_params(B))
        return self._v("zh", self.pad16(B), self.spec.zin)[:B, :self.spec.Z]

    def _encode(self, B):
        self._have_targets = False
        if self._weights_dirty:
            self.prepare_weights()
        else:
            self._zero_scal()
        self.encoder_forward(B)
        self._verify_pipeline(lambda: (self.scal.zero_(), self.encoder_forward(B)), key="encode")

    def decode(self, B, want_probs=True):
        """``decoder.predict`` on the staged [z|history]; argmax note indices are always produced on device."""
        self._planned(("decode", B, float(self.norm_B), bool(want_probs)), lambda: self._decode(B, want_probs), params=self._call_params(B))

    def _decode(self, B, want_probs):
        self._have_targets = False
        self._S_done = False
        if self._weights_dirty:
            self.prepare_weights()
        else:
            self._zero_scal()
        self.decoder_forward(B, want_probs=want_probs)
        self._verify_pipeline(lambda: (self.scal.zero_(), self.decoder_forward(B, want_probs=want_probs)), key="decode")
