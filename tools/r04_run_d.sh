#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04d
( time python -m pytest tests -x -q -m gpu --durations=25 ) > gpurun_out/r04d/pytest_gpu.txt 2>&1
tail -45 gpurun_out/r04d/pytest_gpu.txt
for args in "" "--with-prepass" "--windows 256 --songs 16" "--windows 256 --songs 16 --with-prepass" "--lazy"; do
  echo "## fit_e2e_bench $args" | tee -a gpurun_out/r04d/fit_e2e.txt
  python tools/fit_e2e_bench.py $args 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04d/fit_e2e.txt
  echo "## fit_e2e_bench $args (MVAE_PLANS=0)" | tee -a gpurun_out/r04d/fit_e2e.txt
  MVAE_PLANS=0 python tools/fit_e2e_bench.py $args 2>&1 | grep -v amdgpu.ids | grep "epoch 3\|host time" | tail -2 | tee -a gpurun_out/r04d/fit_e2e.txt
done
