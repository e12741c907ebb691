#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04g
python -m pytest tests/test_golden_gpu.py tests/test_ops_gpu.py tests/test_baseline_configs_gpu.py -x -q 2>&1 | tail -8
python tools/rnn_microbench.py --cell LSTM --reps 8 2>&1 | grep -v amdgpu | tee gpurun_out/r04g/rnn_microbench_lstm.txt
python tools/decode_bench.py 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r04g/decode.txt
