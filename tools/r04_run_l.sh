#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04l
python -m pytest tests/test_ops_gpu.py -x -q -k "rnn_forward or paired" 2>&1 | tail -3
python tools/rnn_microbench.py --cell GRU --reps 8 2>&1 | grep "fwd" | tee gpurun_out/r04l/gru_fwd_index_paired.txt
for c in GRU LSTM; do for e in "MVAE_INDEX_DENSE=0" "MVAE_INDEX_DENSE=1"; do
  echo "## $c $e" | tee -a gpurun_out/r04l/index_dense_ab.txt
  env $e python bench.py --cell $c --no-cpu-baseline --no-other-configs --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step %.3f value %.0f' % (d['ms_per_step'], d['value']))" | tee -a gpurun_out/r04l/index_dense_ab.txt
done; done
python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_classifier_gpu.py -x -q 2>&1 | tail -3
