cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o /tmp/anyorder_probe && timeout 60 /tmp/anyorder_probe > $O/anyorder_probe.txt 2>&1
cat $O/anyorder_probe.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py -m gpu -q --timeout 600 -x > $O/pytest_subset.txt 2>&1
tail -5 $O/pytest_subset.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['value']), d.get('elbo',{}).get('max_abs_diff'))"; }
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>$O/bench.err | tee $O/bench_lstm_$i.json | line "LSTM"; done
timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/bench.err | tee $O/bench_gru.json | line "GRU"
timeout 900 python bench.py --steps 20 2>>$O/bench.err | tee $O/bench_lstm_full.json | line "LSTM full (cpu+elbo)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/bench_lstm_kernel_stats.csv
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_lstm_step.txt
timeout 600 python bench.py --no-cpu-baseline --config 2 --steps 10 --warmup 3 2>>$O/bench.err | tee $O/bench_cfg2.json | line "cfg2"
timeout 600 python bench.py --no-cpu-baseline --config 4 --steps 10 --warmup 3 2>>$O/bench.err | tee $O/bench_cfg4.json | line "cfg4"
timeout 600 python tools/fit_e2e_bench.py --songs 4 --windows 1024 --with-prepass > $O/fit_e2e_1024_prepass.txt 2>&1; tail -4 $O/fit_e2e_1024_prepass.txt
timeout 600 python tools/fit_e2e_bench.py --songs 4 --windows 1024 > $O/fit_e2e_1024.txt 2>&1; tail -4 $O/fit_e2e_1024.txt
