cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3))"; }
for v in 1 0 1 0 1; do
  MVAE_KSTREAM_SINGLES=$v timeout 300 python bench.py --no-cpu-baseline --cell GRU 2>>$O/gru8.err | line "GRU singles=$v" >> $O/ab_gru8.txt
done
timeout 300 python bench.py --no-cpu-baseline 2>>$O/gru8.err | line "LSTM" >> $O/ab_gru8.txt
cat $O/ab_gru8.txt
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 400 --maxfail 5 > $O/pytest_gru8.txt 2>&1
tail -3 $O/pytest_gru8.txt
