#!/usr/bin/env python3
"""Which neighbours slow a recurrence phase inside the train step?  The four phase launches of a step (encoder / decoder, forward /
BPTT), HIP-event timed, for the full model and with the side rolls / heads removed from the graph (meta_velocity, meta_instrument
off): the notes stack alone in its phase against the notes stack beside the velocity / instrument recurrences.
   python tools/in_step_probe.py [--cell LSTM]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_vae_amd  # noqa
from midi_vae_amd.engine import Engine
from midi_vae_amd.layout import ModelSpec
from midi_vae_amd.synth import make_windows
ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="LSTM")
a = ap.parse_args()
T, B = 512, 256
KEYS = {("rnn_fwd_multi", "enc"), ("rnn_fwd_multi", "dec"), ("rnn_bwd_multi", "dec"), ("rnn_bwd_multi", "enc")}
for name, kw in (("full model", {}), ("no velocity roll / head", dict(meta_velocity=False)),
                 ("no velocity, no instrument", dict(meta_velocity=False, meta_instrument=False)),
                 ("full model, no parameter gradients (timing only)", dict(_env=dict(MVAE_DIAG_NO_PARAM_GRADS="1", MVAE_KSTREAM_GRADS="0")))):
    env = kw.pop("_env", {})
    os.environ.update(env)
    spec = ModelSpec(cell=a.cell, H=256, Z=64, Din=61, Dout=61, T=T, V=4, ID=16, C=2, Le=2, Ld=2, **kw)
    eng = Engine(spec, max_batch=B, dtype="bf16", device="cuda:0", seed=1)
    for k in env:
        os.environ.pop(k)
    w = make_windows(B, T, 61, 4, 16, 2, 64, seed=1, epsilon_std=spec.epsilon_std)
    eng.stage_encoder_inputs(w["x_idx"], w["i_idx"] if spec.meta_instrument else None, w["vel"] if spec.meta_velocity else None, w["eps"])
    eng.stage_decoder_inputs(B, hist=w["hist"])
    eng.stage_targets(B, w["x_idx"], w["c_idx"])
    for _ in range(20):
        eng.train_step(B)
    torch.cuda.synchronize()
    eng.prof_kinds, eng.prof = KEYS, {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.train_step(B)
    e1.record()
    torch.cuda.synchronize()
    s = eng.prof_summary()
    eng.prof = None
    eng.check_pipeline()
    print("%-52s step %.3f ms | " % (name, e0.elapsed_time(e1) / 10) +
          "  ".join("%s:%s %.3f ms (%.2f us/step)" % (k[0][4:7], k[1], v[1], v[1] * 1e3 / T) for k, v in sorted(s.items())))
    del eng
    torch.cuda.empty_cache()
