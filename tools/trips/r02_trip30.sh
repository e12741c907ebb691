cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py tests/test_dp_fit_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 400 --maxfail 5 > $O/pytest_ks4.txt 2>&1
tail -5 $O/pytest_ks4.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "kstream enc+dec LSTM" >> $O/ab_ks4.txt; done
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "kstream enc+dec GRU" >> $O/ab_ks4.txt; done
cat $O/ab_ks4.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_ks4.txt
