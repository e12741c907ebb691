cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O; cd $R
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_dec -- python bench.py --config 4 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_config4_traced.json 2>/dev/null
cp $(find /tmp/ks_dec -name "*kernel_stats.csv" | head -1) $O/bench_config4_LSTM_kernel_stats.csv
head -8 $O/bench_config4_LSTM_kernel_stats.csv | cut -c1-160
