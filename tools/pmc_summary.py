#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs of tools/rnn_microbench.py (T=512, B=256, H=256 bf16) per
recurrent kernel: mean per launch, per-wave-per-step instruction counts, MFMA-busy fraction, HBM bytes.
   python tools/pmc_summary.py gpurun_out/prof_c/pmc_*_counter_collection.csv > profiles/<round>_rnn_pmc_summary.txt"""
import collections, csv, re, sys
T, B = 512, 256
agg = collections.defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void |\(mvae_rnn_\w+_args\)", "", r["Kernel_Name"])
        if "lstm" in k or "rnn_" in k or "gru_" in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
kernels = sorted({k for k, _ in agg})
m = lambda k, c: (sum(agg[(k, c)]) / len(agg[(k, c)])) if (k, c) in agg else float("nan")
# writes: WRITE_SIZE (KiB) or - its pass hangs in rocprofv3's start-up on this image - TCC_EA0_WRREQ_sum x 64 B
written_mb = lambda k: m(k, "WRITE_SIZE") * 1024 / 1e6 if (k, "WRITE_SIZE") in agg else m(k, "TCC_EA0_WRREQ_sum") * 64 / 1e6
print("rocprofv3 --pmc, one pass per counter group; tools/rnn_microbench.py --cell LSTM / GRU (T=%d steps, B=%d rows, H=256, bf16)" % (T, B))
print("per launch: %d workgroups x 4 waves (x 8 for the two-waves-per-SIMD *_w8_k kernels); SQ_*_CYCLES counters are quad-cycles except "
      "SQ_VALU_MFMA_BUSY_CYCLES (cycles)" % (B // 16))
print("FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950\n")
for k in kernels:
    waves = (B // 16) * (8 if "_w8" in k else 4)
    wc, mf = m(k, "SQ_WAVE_CYCLES"), m(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    print(k)
    print("  cycles per wave per time step      %8.0f   (= %.2f us at 2.4 GHz)" % (wc * 4 / waves / T, wc * 4 / waves / T / 2400))
    print("  MFMA pipe busy                     %8.1f %% of wave cycles" % (100 * mf / (wc * 4)))
    print("  issue: active %4.1f %%  wait-inst %4.1f %%  wait-any (s_waitcnt/barrier) %4.1f %%" % (
        100 * m(k, "SQ_ACTIVE_INST_ANY") / wc, 100 * m(k, "SQ_WAIT_INST_ANY") / wc, 100 * m(k, "SQ_WAIT_ANY") / wc))
    # SQ_INSTS_VALU counts the MFMAs too (they are VALU-encoded): the other vector instructions are the difference
    n_mfma, n_valu, n_lds, n_vmem = (m(k, c) / waves / T for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"))
    print("  per wave per step: MFMA %.0f  other VALU %.0f (SQ_INSTS_VALU %.0f includes the MFMAs)  LDS %.0f  VMEM %.0f" % (
        n_mfma, n_valu - n_mfma, n_valu, n_lds, n_vmem))
    if (k, "FETCH_SIZE") in agg or (k, "WRITE_SIZE") in agg or (k, "TCC_EA0_WRREQ_sum") in agg:     # (their own passes: tools/pmc_traffic.py)
        print("  HBM per launch: read %.1f MB (2 x FETCH_SIZE)  written %.1f MB" % (
            2 * m(k, "FETCH_SIZE") * 1024 / 1e6, written_mb(k)))
