#!/bin/bash
# experiment F: what slows the recurrences inside the step?  (dX GEMM grid, K-streaming, all parameter-gradient work off)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04f
run() { echo "## $*" | tee -a gpurun_out/r04f/in_step.txt; env "$@" python bench.py --no-cpu-baseline --no-other-configs --steps 24 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  bwd us/step %.3f  fwd us/step %.3f  launches %s' % (d['ms_per_step'], r['us_per_time_step'], r['critical_path']['us_per_step_fwd'], {k: round(v,3) for k,v in r['launch_ms_by_layer'].items()}))" | tee -a gpurun_out/r04f/in_step.txt; }
run MVAE_X=0
run MVAE_PIPE_GEMM_BLOCKS=32
run MVAE_PIPE_GEMM_BLOCKS=16
run MVAE_KSTREAM_GRADS=0
run MVAE_DIAG_NO_PARAM_GRADS=1 MVAE_KSTREAM_GRADS=0
run MVAE_DIAG_NO_PARAM_GRADS=1 MVAE_KSTREAM_GRADS=0 MVAE_PIPE_GEMM_BLOCKS=16
