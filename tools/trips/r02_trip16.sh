cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py -m gpu -q --timeout 400 --maxfail 5 -x -k "pipelin or resident or baseline or classifier or hand_over" > $O/pytest_wt2.txt 2>&1
tail -5 $O/pytest_wt2.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 32 16 64 8 32 16 64; do
  MVAE_PIPE_CHUNK=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "wt pipe_chunk=$v LSTM" >> $O/ab_wt2.txt
done
for v in 32 16; do
  MVAE_PIPE_CHUNK=$v timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "wt pipe_chunk=$v GRU" >> $O/ab_wt2.txt
done
cat $O/ab_wt2.txt
