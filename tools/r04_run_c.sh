#!/bin/bash
# GPU run C: plan tests, host-vs-device with plans, dhs A/B, bwd kernel parity
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04c
python -m pytest tests/test_plan_gpu.py -x -q 2>&1 | tail -25
python -m pytest tests/test_ops_gpu.py -x -q -k "rnn_backward" 2>&1 | tail -3
for shape in bench reference; do python tools/plan_host_bench.py --shape $shape 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04c/plan_host_bench.txt; done
python tools/plan_host_bench.py --shape reference --cell LSTM 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04c/plan_host_bench.txt
for v in olddhs product; do
  echo "## $v" | tee -a gpurun_out/r04c/dhs_ab.txt
  if [ $v = product ]; then unset MVAE_LIB; else export MVAE_LIB=$PWD/build/variants/lib_$v.so; fi
  python tools/rnn_microbench.py --cell LSTM --reps 8 2>&1 | grep "bwd" | tee -a gpurun_out/r04c/dhs_ab.txt
  python tools/rnn_microbench.py --cell LSTM --reps 8 2>&1 | grep "bwd" | tee -a gpurun_out/r04c/dhs_ab.txt
done
