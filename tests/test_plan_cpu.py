"""Step plans without a GPU: the C-side container (csrc/plan.cpp) validates what it is given, and plan.StepPlan turns three
recordings into constants + counter patches - or refuses (include/midivae_hip.h 'STEP PLANS').  No entry point is executed here:
running a plan needs the device (tests/test_plan_gpu.py)."""
import ctypes as C

import pytest

import midi_vae_amd  # noqa: F401
from midi_vae_amd import hiplib as hl
from midi_vae_amd import plan as P


def _new():
    lib = hl.load()
    h = C.c_void_p()
    assert lib.mvae_plan_create(C.byref(h)) == 0
    return lib, h


def test_container_validates_names_arity_blobs_and_patches():
    lib, h = _new()
    slots = (C.c_uint64 * 3)(0, 0, 7)
    assert lib.mvae_plan_add_call(h, b"mvae_no_such_entry", slots, 3) == hl.E_ARG
    assert lib.mvae_plan_add_call(h, b"mvae_stream_wait_value32", slots, 2) == hl.E_ARG          # three arguments, not two
    assert lib.mvae_plan_add_call(h, b"mvae_host_threads", slots, 1) == hl.E_ARG                 # not a stream-taking entry point
    assert lib.mvae_plan_add_call(h, b"mvae_stream_wait_value32", slots, 3) == 0
    two = (C.c_uint64 * 2)(0, 0)
    assert lib.mvae_plan_add_call(h, b"mvae_gemm", two, 2) == 1
    assert lib.mvae_plan_size(h) == 2
    g = hl.GemmArgs()
    assert lib.mvae_plan_set_blob(h, 1, 0, C.addressof(g), C.sizeof(g)) == 0
    assert lib.mvae_plan_set_blob(h, 1, 0, C.addressof(g), C.sizeof(g)) == hl.E_ARG             # one blob per argument
    assert lib.mvae_plan_set_blob(h, 5, 0, C.addressof(g), C.sizeof(g)) == hl.E_ARG
    off = hl.GemmArgs.chunk_wait_value.offset
    assert lib.mvae_plan_add_patch(h, 1, 0, off, 0, 0) == 0
    assert lib.mvae_plan_add_patch(h, 1, 0, C.sizeof(g), 0, 0) == hl.E_ARG                       # beyond the struct
    assert lib.mvae_plan_add_patch(h, 1, 0, off + 2, 0, 0) == hl.E_ARG                           # not a 32-bit word
    assert lib.mvae_plan_add_patch(h, 0, 2, -1, 1, 4) == 0                                       # a scalar argument
    # running: the patches need their key values; a rejected call stops the run and is reported
    assert lib.mvae_plan_run(h, 0, -1, None, 0) == hl.E_ARG
    keys = (C.c_uint64 * 2)(10, 20)
    assert lib.mvae_plan_run(h, 0, -1, keys, 2) == hl.E_ARG        # mvae_stream_wait_value32(NULL address) rejects: nothing is enqueued
    assert lib.mvae_plan_failed_call(h) == 0
    assert lib.mvae_plan_run(h, 1, 2, keys, 2) == hl.E_ARG and lib.mvae_plan_failed_call(h) == 1     # (an all-zero mvae_gemm_args)
    assert lib.mvae_plan_run(h, 2, 2, keys, 2) == 0 and lib.mvae_plan_failed_call(h) == -1           # an empty range
    assert lib.mvae_plan_destroy(h) == 0


def _recording(base, seq):
    """the calls a step with counter ("sync", 0, 0) = base (before the step) and join sequence seq would make"""
    g = hl.GemmArgs(M=128, N=64, K=32, A=0x1000, B=0x2000, C=0x3000, chunk_wait=0x4000, chunk_wait_value=base + 64)
    calls = [("mvae_gemm", [0, 0x77], {0: C.string_at(C.addressof(g), C.sizeof(g))}),
             ("mvae_stream_write_value32", [0x77, 0x5000, seq + 1], {}),
             ("mvae_stream_wait_value32", [0x78, 0x5000, seq + 1], {}),
             ("mvae_adam_step_dev", P._encode(hl.SIGNATURES["mvae_adam_step_dev"][1],
                                              (1, 2, 3, 4, 100, 2e-4, 0.9, 0.999, 1e-8, 5, 1.0, 3, 6, 0x77))[0], {})]
    tags = {(0, 0, hl.GemmArgs.chunk_wait_value.offset): ("sync", 0, 0), (1, 2, -1): ("join", 2), (2, 2, -1): ("join", 2)}
    before = {("sync", 0, 0): base, ("join", 2): seq}
    after = {("sync", 0, 0): base + 64, ("join", 2): seq + 1}
    return calls, tags, before, after


def test_three_recordings_become_constants_and_counter_patches():
    # (the recordings need not be consecutive steps: the patches are relative to the counters, not to a run index; two counters
    #  that happen to hold the same value are no problem either: a field is tied to its counter by name, not by value)
    p = P.StepPlan([_recording(0, 0), _recording(64, 1), _recording(640, 17)])
    assert p.n_calls == 4 and p.n_patches == 3
    assert p.inc == {("sync", 0, 0): 64, ("join", 2): 1}
    p.close()


def test_floats_travel_as_their_bits():
    slots, _ = P._encode([C.c_void_p, C.c_float, C.c_int32], (None, 0.5, -1))
    assert slots == [0, 0x3F000000, (1 << 64) - 1]


@pytest.mark.parametrize("what", ["pointer", "untagged", "length", "offset"])
def test_anything_else_that_differs_is_refused(what):
    a, b, c = _recording(0, 0), _recording(64, 1), _recording(128, 2)
    if what == "pointer":            # a buffer that moved between steps
        c[0][1] = ("mvae_stream_write_value32", [0x77, 0x9000_0000_0000, 3], {})
    elif what == "untagged":         # a 32-bit value that changes but was never announced as a counter value
        c[0][3][1][4] = 101
    elif what == "length":
        c[0].pop()
    else:                            # the same counter, but at another distance from its value before the step
        g = hl.GemmArgs(M=128, N=64, K=32, A=0x1000, B=0x2000, C=0x3000, chunk_wait=0x4000, chunk_wait_value=128 + 65)
        c[0][0] = ("mvae_gemm", [0, 0x77], {0: C.string_at(C.addressof(g), C.sizeof(g))})
    with pytest.raises(P.NotReplayable):
        P.StepPlan([a, b, c])


def test_recorder_notes_accepted_calls_only_and_restores_the_library():
    lib = hl.load()
    real = lib.mvae_gemm
    with P.Recorder() as rec:
        assert lib.mvae_gemm is not real
        assert lib.mvae_gemm(None, None) == hl.E_ARG          # rejected: enqueued nothing, not part of the step
        rec.note_field(0, 8, ("sync", 1, 0))                  # announced for a call that was then rejected: dropped with it
        assert lib.mvae_gemm(None, None) == hl.E_ARG
    assert lib.mvae_gemm is real and rec.calls == [] and rec.tags == {} and rec.tainted is None
    assert P.active() is None


def test_counter_values_announce_the_fields_they_land_in():
    """ops.CounterValue -> the marshalling layer tags the struct field -> the recorder learns (call, argument, byte offset)"""
    from midi_vae_amd import ops
    v = ops.CounterValue(192, ("sync", 3, 1))
    assert int(v) == 192 and v.key == ("sync", 3, 1) and not hasattr(v + 1, "key")
    g = hl.GemmArgs()
    ops._tag(g, "k_wait_value", v)
    ops._tag(g, "chunk_wait_value", 5)                       # a plain int: not a counter
    assert g._counter_fields == {"k_wait_value": ("sync", 3, 1)}
    g2 = hl.GemmArgs()
    with P.Recorder() as rec:
        ops._note_fields(0, [g2, g])
        assert rec._pending == [(0, C.sizeof(hl.GemmArgs) + hl.GemmArgs.k_wait_value.offset, ("sync", 3, 1))]
        ops._note_scalar(2, v)
        assert rec._pending[-1] == (2, -1, ("sync", 3, 1))


def test_host_marks_must_agree_and_split_the_replay_into_ranges():
    """A data-parallel step's collectives are HOST actions between two launches (Recorder.host): the three recordings must hold them
    at the same places, and a replay calls them between the call ranges (run_ranges) - checked here on ranges no call of which
    can run without a device, so only the order of (range, host action) is observed."""
    a, b, c = _recording(0, 0), _recording(64, 1), _recording(128, 2)
    m = [(1, "early", 0x99), (3, "reduce", 0x77), (3, "status", 0x77)]
    with pytest.raises(P.NotReplayable):
        P.StepPlan([a, b, c], [m, m, m[:2]])
    with pytest.raises(P.NotReplayable):
        P.StepPlan([a, b, c], [m, m, [(2, "early", 0x99)] + m[1:]])
    p = P.StepPlan([a, b, c], [m, m, m])
    assert p.marks == m
    seen = []
    real_run = p.run
    p.run = lambda counters, first=0, last=-1: seen.append(("calls", first, last)) or {}
    p.run_ranges({("sync", 0, 0): 0, ("join", 2): 0}, lambda tag, st: seen.append((tag, st)))
    assert seen == [("calls", 0, 1), ("early", 0x99), ("calls", 1, 3), ("reduce", 0x77), ("status", 0x77), ("calls", 3, -1)]
    p.run = real_run
    p.close()


def test_recorder_host_runs_the_action_outside_the_taint_mode_and_notes_the_mark():
    import torch
    with P.Recorder() as rec:
        out = rec.host("reduce", 0x55, lambda: float(torch.ones(3).sum()))        # a torch kernel inside a host action: no taint
        assert out == 3.0 and rec.tainted is None
        torch.ones(2) + 1                                                           # ... outside one: the step is not replayable
    assert rec.tainted is not None and rec.marks == [(0, "reduce", 0x55)]


def test_call_parameters_are_patched_whether_or_not_the_recordings_differ():
    """ops.ParamInt / ParamFloat (key ("param", name)): the real number of windows of a padded minibatch and 1 / global minibatch
    size reach a few launches as tagged values; the plan patches every such field with the value the caller names at replay - also
    when the three recordings happened to carry the same value (songs of equal length) - so steps on 97 and 100 windows share one
    plan.  A recording whose caller named no value for a parameter it used is refused."""
    from midi_vae_amd import ops
    recs = []
    for base, seq, nwin in ((0, 0, 100), (64, 1, 100), (128, 2, 100)):
        calls, tags, before, after = _recording(base, seq)
        g = hl.GemmArgs(M=128, N=64, K=32, A=0x1000, B=0x2000, C=0x3000, chunk_wait=0x4000, chunk_wait_value=base + 64, alpha=1.0 / nwin)
        calls[0] = ("mvae_gemm", [0, 0x77], {0: C.string_at(C.addressof(g), C.sizeof(g))})
        calls[3][1][4] = nwin                                               # (an integer argument that carries the window count)
        tags = dict(tags)
        tags[(3, 4, -1)] = ops.PARAM_B
        tags[(0, 0, hl.GemmArgs.alpha.offset)] = ops.PARAM_INV_BATCH
        before = dict(before)
        before[ops.PARAM_B], before[ops.PARAM_INV_BATCH] = nwin, ops.f32_bits(1.0 / nwin)
        recs.append((calls, tags, before, after))
    p = P.StepPlan(recs)
    assert p.params == sorted([ops.PARAM_B, ops.PARAM_INV_BATCH], key=repr) and p.n_patches == 3 + 2
    assert set(p.keys) == {("sync", 0, 0), ("join", 2), ops.PARAM_B, ops.PARAM_INV_BATCH} and set(p.inc) == {("sync", 0, 0), ("join", 2)}
    p.close()
    missing = [(c, t, {k: v for k, v in b.items() if k != ops.PARAM_B}, a) for c, t, b, a in recs]
    with pytest.raises(P.NotReplayable):
        P.StepPlan(missing)
    assert ops.f32_bits(0.5) == 0x3F000000 and ops.ParamInt(7, ops.PARAM_B).key == ops.PARAM_B and float(ops.ParamFloat(0.25, ops.PARAM_INV_BATCH)) == 0.25
