#!/bin/bash
# Round 6: issue / MFMA counters of the GRU recurrent kernels alone - the 4-wave slot-interleaved kernels and the two-waves-per-SIMD
# ones - one rocprofv3 --pmc pass per counter group (never combined with a trace domain).   gpurun -- 'bash tools/collect_pmc_gru_r06.sh'
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/pmc_r06; mkdir -p $O
export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS")
for fam in w8 il; do
  i=0
  for g in "${GROUPS_[@]}"; do
    i=$((i+1)); rm -rf /tmp/pmc_${fam}_$i
    ( cd /tmp && timeout -k 5 170 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_${fam}_$i -- python $OLDPWD/tools/rnn_microbench.py --cell GRU $([ $fam = w8 ] && echo --w8) > /dev/null 2>&1 )
  done
  python tools/pmc_summary.py $(find /tmp/pmc_${fam}_* -name "*counter_collection.csv") > $O/gru_${fam}_pmc_summary.txt 2>&1
done
tail -n 60 $O/*.txt
