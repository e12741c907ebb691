cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 4 1 2 8 4 1; do
  MVAE_SINGLE_BWD_CHUNKS=$v timeout 600 python bench.py --no-cpu-baseline 2>>$O/sc.err | line "single_bwd_chunks=$v LSTM" >> $O/ab_sc.txt
done
for v in 4 1 2; do
  MVAE_SINGLE_BWD_CHUNKS=$v timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/sc.err | line "single_bwd_chunks=$v GRU" >> $O/ab_sc.txt
done
cat $O/ab_sc.txt
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py -m gpu -q --timeout 400 --maxfail 5 > $O/pytest_sc.txt 2>&1
tail -4 $O/pytest_sc.txt
