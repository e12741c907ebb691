#!/usr/bin/env python3
"""Reconstruct one training step's kernel timeline from a rocprofv3 --kernel-trace CSV.
   python tools/timeline.py <kernel_trace.csv> [--step -2] [--min-us 30]
Steps are delimited by the optimizer kernel (adam / rmsprop).  Prints every kernel of the chosen step that runs longer
than --min-us (start, duration, queue), the busy time per queue and the union busy time."""
import argparse, csv, re, sys
ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--step", type=int, default=-2); ap.add_argument("--min-us", type=float, default=30.0)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if re.search(r"adam|rmsprop", r[2])]
if len(ends) < 3:
    sys.exit("fewer than 3 optimizer launches in the trace")
lo, hi = ends[a.step - 1] + 1, ends[a.step] + 1
step = rows[lo:hi]
t0 = rows[lo - 1][1]                      # end of the previous optimizer kernel
print("step: %d kernels, %.3f ms from previous optimizer end to this optimizer end" % (len(step), (step[-1][1] - t0) / 1e6))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:70]
busy = {}
for s, e, n, q in step:
    busy[q] = busy.get(q, 0) + (e - s)
    if (e - s) / 1e3 >= a.min_us:
        print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short(n)))
print("busy per queue (ms):", {q: round(v / 1e6, 3) for q, v in sorted(busy.items())})
iv = sorted((s, e) for s, e, _, _ in step)
u, cs, ce = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        u += ce - cs; gaps.append((cs and ce, s - ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print("union busy %.3f ms; idle gaps > 20 us:" % (u / 1e6), [(round((g0 - t0) / 1e3, 1), round(d / 1e3, 1)) for g0, d in gaps if d > 20e3])
