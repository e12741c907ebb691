#!/bin/bash
# GPU side of experiment B: wave skew variants
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04b
for v in base skew1 skew2 skew3 skew4 skew6; do
  echo "## $v" >> gpurun_out/r04b/wave_skew.txt
  MVAE_LIB=$PWD/build/variants/lib_$v.so timeout 120 python tools/rnn_microbench.py --cell LSTM --reps 5 2>&1 | grep -v "^$" >> gpurun_out/r04b/wave_skew.txt
done
grep "^##\|bwd\|fwd dense\|fwd const " gpurun_out/r04b/wave_skew.txt
