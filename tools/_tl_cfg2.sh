#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04m
cd $R
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_c2 -- python bench.py --config 2 --no-cpu-baseline --no-other-configs --steps 4 --warmup 2 --prewarm-max 1 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_c2 -name "*kernel_trace.csv" | head -1) --min-us 100 > $R/gpurun_out/r04m/timeline_config2_lstm.txt
cat $R/gpurun_out/r04m/timeline_config2_lstm.txt | cut -c1-150
