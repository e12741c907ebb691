#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04k
python -m pytest tests/test_ops_gpu.py -x -q -k "rnn_forward or paired or prepare" 2>&1 | tail -5
python tools/rnn_microbench.py --cell LSTM --reps 8 2>&1 | grep "fwd" | tee gpurun_out/r04k/fwd_index_paired.txt
for v in nb1 nb2 nb12; do echo "## $v" | tee -a gpurun_out/r04k/bptt_single_barriers.txt; MVAE_LIB=$PWD/build/variants/lib_$v.so python tools/rnn_microbench.py --cell LSTM --reps 8 2>&1 | grep "bwd" | tee -a gpurun_out/r04k/bptt_single_barriers.txt; done
python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py -x -q 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-other-configs --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step %.3f value %.0f' % (d['ms_per_step'], d['value']))"
