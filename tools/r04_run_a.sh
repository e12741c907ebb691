#!/bin/bash
# GPU side of experiment A: rnn_microbench for every variant library (LSTM; bwd lines are what matter)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04a
for v in base nol nox notrg nobar nob nomath skel skelnm fill1 fill2 fill3 fill4; do
  echo "## $v" >> gpurun_out/r04a/rnn_ablation.txt
  MVAE_LIB=$PWD/build/variants/lib_$v.so timeout 120 python tools/rnn_microbench.py --cell LSTM --reps 5 2>&1 | grep -v "^$" >> gpurun_out/r04a/rnn_ablation.txt
done
cat gpurun_out/r04a/rnn_ablation.txt
