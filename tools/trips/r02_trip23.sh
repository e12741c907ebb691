# K-streaming weight-gradient GEMMs behind the encoder BPTT: tests, A/B, timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 200 --maxfail 5 -k "gemm" > $O/pytest_ks2_ops.txt 2>&1
tail -5 $O/pytest_ks2_ops.txt
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py tests/test_dp_fit_gpu.py -m gpu -q --timeout 400 --maxfail 5 -x > $O/pytest_ks2.txt 2>&1
tail -5 $O/pytest_ks2.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in "1 32" "0 32" "1 64" "1 48" "1 32" "0 32"; do
  set -- $v
  MVAE_KSTREAM_GRADS=$1 MVAE_KSTREAM_WGS=$2 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "kstream=$1 wgs=$2 LSTM" >> $O/ab_ks2.txt
done
for v in "1 32" "0 32" "1 64"; do
  set -- $v
  MVAE_KSTREAM_GRADS=$1 MVAE_KSTREAM_WGS=$2 timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "kstream=$1 wgs=$2 GRU" >> $O/ab_ks2.txt
done
cat $O/ab_ks2.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_ks2.txt
