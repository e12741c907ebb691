# gated gradient portions of the encoder stack: tests, A/B, timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py tests/test_dp_fit_gpu.py -m gpu -q --timeout 400 --maxfail 5 -x > $O/pytest_gated.txt 2>&1
tail -5 $O/pytest_gated.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 1 0 1 0; do
  MVAE_GATED_GRADS=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "gated=$v LSTM" >> $O/ab_gated.txt
done
MVAE_GATED_GRADS=1 MVAE_GATED_LAST=1 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "gated=1 last=1 LSTM" >> $O/ab_gated.txt
MVAE_GATED_GRADS=1 MVAE_GATED_LAST=4 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "gated=1 last=4 LSTM" >> $O/ab_gated.txt
for v in 1 0; do
  MVAE_GATED_GRADS=$v timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "gated=$v GRU" >> $O/ab_gated.txt
done
cat $O/ab_gated.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_gated.txt
