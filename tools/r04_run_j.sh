#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04j
O=gpurun_out/r04j/bptt_in_step_factors.txt
for v in xa xb xc xd xe; do
  echo "## variant $v" | tee -a $O
  MVAE_LIB=$PWD/build/variants/lib_$v.so python tools/rnn_microbench.py --cell LSTM --reps 8 2>&1 | grep "bwd\|fwd dense" | tee -a $O
done
for args in "" "--signal 16" "--concurrent 3" "--concurrent 3 --signal 16" "--concurrent 2" "--concurrent 6"; do
  echo "## product $args" | tee -a $O
  python tools/rnn_microbench.py --cell LSTM --reps 8 $args 2>&1 | grep "bwd\|fwd dense\|fwd const " | tee -a $O
done
