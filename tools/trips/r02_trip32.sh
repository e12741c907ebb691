cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline 2>>$O/ks5.err | line "kstream enc gated LSTM" >> $O/ab_ks5.txt; done
for i in 1 2 3 4; do timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/ks5.err | line "kstream enc gated GRU" >> $O/ab_ks5.txt; done
cat $O/ab_ks5.txt
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py tests/test_dp_fit_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 400 --maxfail 5 > $O/pytest_ks5.txt 2>&1
tail -5 $O/pytest_ks5.txt
