"""midi-vae_amd: MI355X-native engine for the MIDI-VAE train / inference step.

The directory name carries a hyphen (it mirrors the reference repo's name), so it is imported
through the repo-root shim ``midi_vae_amd.py`` which registers this directory as the package
``midi_vae_amd``.  Sub-modules:

  config    settings surface (reference settings.py names)
  packers   host packers + argmax decode (reference vae_definition.py:770-1235)
  layout    parameter layout of the model (names, shapes, offsets in the flat fp32 buffer)
  hiplib    ctypes binding of the C-ABI library built from csrc/ (include/midivae_hip.h)
  engine    device engine: resident buffers, train step, encode / decode
  model     ``VAE`` facade with the reference's ``create`` / ``fit`` / ``evaluate`` / ``predict`` surface
  synth     synthetic piano-roll windows
  dp        data-parallel sharding over torch.distributed (RCCL)
"""
import os as _os

# The engine runs the independent branches of the step on separate HIP streams (notes stack / velocity / instrument /
# pipelined layers / gradient GEMMs).  ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
# with more streams than queues, unrelated streams share a queue and serialise (seen in profiles/r01_b: the
# gradient GEMMs delayed the lower layer's BPTT by milliseconds).  Must be set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

# One process per GPU (torch.distributed.run sets LOCAL_WORLD_SIZE): the host's threads are shared between the ranks of the node.
# Without a limit each rank's OpenMP / torch intra-op pool is sized for the whole host (8 x 256 threads); the packer pool of the C
# library applies the same rule (csrc/hostpack.cpp default_threads).  An explicit OMP_NUM_THREADS wins.
_lws = int(_os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
if _lws > 1 and "OMP_NUM_THREADS" not in _os.environ:
    try:
        _cores = len(_os.sched_getaffinity(0))
    except AttributeError:
        _cores = _os.cpu_count() or 1
    _os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, _cores // _lws // 4 or 1)))

__version__ = "0.1.0"
