#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04h
for v in v00 v01 v10 v11 v00 v11; do
  echo "## $v (1MSQ,FIRST)" | tee -a gpurun_out/r04h/bwd_ab.txt
  MVAE_LIB=$PWD/build/variants/lib_$v.so python tools/rnn_microbench.py --cell LSTM --reps 10 2>&1 | grep "bwd" | tee -a gpurun_out/r04h/bwd_ab.txt
done
