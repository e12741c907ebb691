#!/usr/bin/env python3
"""HBM traffic of the dominant kernel (BPTT of the T-step layers) from rocprofv3 PMC passes over bench.py ITSELF.
   python tools/pmc_traffic.py --cell LSTM --dtype bf16 --T 512 --B 256 --fetch <FETCH_SIZE pass counter_collection.csv> \
       --write <WRITE_SIZE pass counter_collection.csv> --out profiles/r02_bench_traffic.json
FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 counter slots: FETCH_SIZE takes 3, WRITE_SIZE 2).  Units: KiB.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte read requests of wide coalesced loads at
64 bytes - doubled here; WRITE_SIZE counted 64-byte write requests exactly in the calibration below.
Calibration on this kernel (profiles/r02_pmc_counters.txt): TCC_EA0_RDREQ_sum x 128 B = 2 x FETCH_SIZE and TCC_EA0_WRREQ_sum x 64 B
= WRITE_SIZE, and both equal the algorithmic bytes of a launch to 1-2 %."""
import argparse, collections, csv, json, os, re

ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="LSTM")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--out", required=True)
a = ap.parse_args()
pat = "lstm_bwd_il_k" if a.cell == "LSTM" else "gru_bwd_il_k"


def per_launch(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if pat in r["Kernel_Name"] and r["Counter_Name"] == counter]
    srt = sorted(vals)
    med_top = srt[int(0.9 * len(srt))]                          # (robust against a single outlier launch)
    big = [v for v in vals if v > 0.3 * med_top]                # the T-step launches (the 4-step instrument layers are 100x smaller)
    print("%s %s: %d launches; deciles %s" % (pat, counter, len(vals), ["%.3g" % srt[int(q * (len(srt) - 1) / 10)] for q in range(11)]))
    return sum(big) / len(big), len(big), len(vals)


f, nf, tf = per_launch(a.fetch, "FETCH_SIZE")
w, nw, tw = per_launch(a.write, "WRITE_SIZE")
rd, wr = 2.0 * f * 1024.0, w * 1024.0
H = 256
alg = a.B * a.T * H * (10 if a.cell == "LSTM" else 9) * (2 if a.dtype == "bf16" else 4)
rec = {"kernel": pat, "T": a.T, "B": a.B, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg, "launches_averaged": [nf, nw],
       "launches_seen": [tf, tw],
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --no-cpu-baseline`; "
                 "2 x FETCH_SIZE KiB + WRITE_SIZE KiB, mean over the T-step launches of the kernel (tools/pmc_traffic.py)"}
out = {}
if os.path.exists(a.out):
    out = json.load(open(a.out))
out["%s_%s" % (a.cell, a.dtype)] = rec
json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
print(json.dumps(rec))
