#!/usr/bin/env python3
"""HBM traffic of the dominant kernel (BPTT of the T-step layers) from rocprofv3 PMC passes over bench.py ITSELF.
   python tools/pmc_traffic.py --cell LSTM --dtype bf16 --T 512 --B 256 --fetch <FETCH_SIZE pass counter_collection.csv> \
       --write <WRITE_SIZE pass counter_collection.csv> --out profiles/r02_bench_traffic.json
FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 counter slots: FETCH_SIZE takes 3, WRITE_SIZE 2).  Units: KiB.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte read requests of wide coalesced loads at
64 bytes - doubled here; WRITE_SIZE counted 64-byte write requests exactly in the calibration below.
Calibration on this kernel (profiles/r02_pmc_counters.txt): TCC_EA0_RDREQ_sum x 128 B = 2 x FETCH_SIZE and TCC_EA0_WRREQ_sum x 64 B
= WRITE_SIZE, and both equal the algorithmic bytes of a launch to 1-2 %."""
import statistics
import argparse, collections, csv, json, os, re

ap = argparse.ArgumentParser()
ap.add_argument("--cell", default="LSTM")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--T", type=int, default=512)
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--out", required=True)
ap.add_argument("--write-counter", default="WRITE_SIZE", choices=["WRITE_SIZE", "TCC_EA0_WRREQ_sum"],
                help="WRITE_SIZE (KiB) or TCC_EA0_WRREQ_sum (64-byte write requests; calibrated equal in round 2 - used in round 3, "
                     "where the WRITE_SIZE pass hangs in rocprofv3's start-up on this image)")
a = ap.parse_args()
pat = "lstm_bwd_il_k" if a.cell == "LSTM" else "gru_bwd_il_k"


def per_launch(path, counter):
    first = open(path).readline()
    if "Counter_Name" in first:     # rocprofv3's counter_collection.csv
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if pat in r["Kernel_Name"] and r["Counter_Name"] == counter]
    else:                           # the kernel's rows as committed under profiles/ (no header): the value follows the counter's name
        vals = []
        for row in csv.reader(open(path)):
            if counter in row and any(pat in c for c in row):
                vals.append(float(row[row.index(counter) + 1]))
    srt = sorted(vals)
    # Launch sizes seen under the profiler: the 4-step instrument layers (100x smaller), 128-step chunks (counter collection runs
    # one kernel at a time, so the time-pipelined stacks - whose kernels wait for producers running BESIDE them - fall back to one
    # launch per layer and chunk: Engine._verify_pipeline), and whole T-step launches (the layers that are not stacked).  The bench
    # brackets T-step launches WITH an upstream-gradient sequence: the largest class.  Encoder layers without one read 1/6 less.
    top = srt[-1] if len(srt) < 20 else srt[int(0.98 * len(srt))]
    big = [v for v in vals if v > 0.93 * top]
    print("%s %s: %d launches; deciles %s; %d whole-sequence launches averaged" % (
        pat, counter, len(vals), ["%.3g" % srt[int(q * (len(srt) - 1) / 10)] for q in range(11)], len(big)))
    # MEDIAN of the class: the launches of the first step WAIT for producers that cannot run beside them under counter collection
    # (before the engine falls back, twice since the first use is retried once) and their polling loads count as fetches
    # (2.3x the bytes of an ordinary launch)
    return statistics.median(big), len(big), len(vals)


f, nf, tf = per_launch(a.fetch, "FETCH_SIZE")
w, nw, tw = per_launch(a.write, a.write_counter)
rd, wr = 2.0 * f * 1024.0, (w * 1024.0 if a.write_counter == "WRITE_SIZE" else w * 64.0)
H = 256
alg = a.B * a.T * H * (10 if a.cell == "LSTM" else 9) * (2 if a.dtype == "bf16" else 4)
rec = {"kernel": pat, "T": a.T, "B": a.B, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg, "launches_averaged": [nf, nw],
       "launches_seen": [tf, tw],
       "write_counter": a.write_counter,
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE or TCC_EA0_WRREQ_sum x 64 B (separate passes) over `python bench.py "
                 "--no-cpu-baseline --steps 10 --warmup 3 --prewarm-max 0`; 2 x FETCH_SIZE KiB + WRITE_SIZE KiB, median over the "
                 "whole-sequence (T-step) launches of the kernel that read an upstream-gradient sequence - under counter collection "
                 "kernels run one at a time, so the stacked layers run as 128-step chunk launches (same bytes per time step) and the "
                 "T-step launches are those of the velocity decoder layer: same kernel, same shape as a stacked decoder layer "
                 "(tools/pmc_traffic.py)"}
out = {}
if os.path.exists(a.out):
    out = json.load(open(a.out))
out["%s_%s" % (a.cell, a.dtype)] = rec
json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
print(json.dumps(rec))
