#!/usr/bin/env python3
"""Mean counter value per launch, per kernel and counter, of a rocprofv3 --pmc counter_collection CSV."""
import collections, csv, re, sys
agg = collections.defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void |\(mvae_\w+_args\)", "", r["Kernel_Name"])[:60]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-60s %-28s n=%-4d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
