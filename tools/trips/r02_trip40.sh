cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 1 0 1 0; do
  MVAE_LATE_G2=$v timeout 600 python bench.py --no-cpu-baseline 2>>$O/lg.err | line "late_g2=$v LSTM" >> $O/ab_lg.txt
done
for v in 1 0; do
  MVAE_LATE_G2=$v timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/lg.err | line "late_g2=$v GRU" >> $O/ab_lg.txt
done
cat $O/ab_lg.txt
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q --timeout 400 --maxfail 5 > $O/pytest_lg.txt 2>&1
tail -4 $O/pytest_lg.txt
