#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc counter_collection CSVs (one pass per counter group) for kernels whose name matches a
pattern: mean per launch of every counter collected, MFMA-busy and issue fractions where their counters are present.
   python tools/pmc_kernels.py --match 'gemm|proj_ws|head_k' <counter_collection.csv ...>"""
import argparse, collections, csv, re
ap = argparse.ArgumentParser()
ap.add_argument("--match", default="gemm|proj_ws")
ap.add_argument("csv", nargs="+")
a = ap.parse_args()
agg = collections.defaultdict(list)
for f in a.csv:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        if re.search(a.match, k):
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
m = lambda k, c: (sum(agg[(k, c)]) / len(agg[(k, c)])) if (k, c) in agg else float("nan")
# writes: WRITE_SIZE (KiB) or - its pass hangs in rocprofv3's start-up on this image - TCC_EA0_WRREQ_sum x 64 B
written_mb = lambda k: m(k, "WRITE_SIZE") * 1024 / 1e6 if (k, "WRITE_SIZE") in agg else m(k, "TCC_EA0_WRREQ_sum") * 64 / 1e6
print("rocprofv3 --pmc, one pass per counter group; means per launch.  SQ_*_CYCLES are quad-cycles summed over waves except")
print("SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES (cycles); FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x 2 on gfx950).\n")
for k in sorted({k for k, _ in agg}):
    n = max(len(v) for (kk, _), v in agg.items() if kk == k)
    wc, mf = m(k, "SQ_WAVE_CYCLES"), m(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    print("%s   (%d launches)" % (k, n))
    if wc == wc:
        print("  MFMA pipe busy %5.1f %% of wave cycles | issue: active %4.1f %%  wait-inst %4.1f %%  wait-any %4.1f %%" % (
            100 * mf / (wc * 4), 100 * m(k, "SQ_ACTIVE_INST_ANY") / wc, 100 * m(k, "SQ_WAIT_INST_ANY") / wc, 100 * m(k, "SQ_WAIT_ANY") / wc))
    print("  instructions per launch: MFMA %.3g  VALU %.3g  LDS %.3g  VMEM %.3g   (VALU per MFMA %.1f)" % (
        m(k, "SQ_INSTS_MFMA"), m(k, "SQ_INSTS_VALU"), m(k, "SQ_INSTS_LDS"), m(k, "SQ_INSTS_VMEM"), m(k, "SQ_INSTS_VALU") / max(m(k, "SQ_INSTS_MFMA"), 1)))
    if (k, "SQ_LDS_BANK_CONFLICT") in agg:
        print("  LDS bank-conflict cycles %.3g of %.3g LDS-active cycles (%.1f %%)" % (
            m(k, "SQ_LDS_BANK_CONFLICT"), m(k, "SQ_LDS_IDX_ACTIVE"), 100 * m(k, "SQ_LDS_BANK_CONFLICT") / max(m(k, "SQ_LDS_IDX_ACTIVE"), 1)))
    if (k, "FETCH_SIZE") in agg or (k, "WRITE_SIZE") in agg:
        print("  HBM per launch: read %.1f MB (2 x FETCH_SIZE)  written %.1f MB" % (2 * m(k, "FETCH_SIZE") * 1024 / 1e6, written_mb(k)))
