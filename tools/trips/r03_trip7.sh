cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches or nonzero_decoder or kstream or optional_heads or three_layer or resident_pipelined or elbo_traj or switches" > $O/pytest_phase.txt 2>&1
tail -8 $O/pytest_phase.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['value']), round(d['roofline']['us_per_time_step'],3), round(d['roofline']['critical_path']['us_per_step_fwd'],3))"; }
for rep in 1 2; do
for m in 1 0; do
  MVAE_PHASE_MULTI=$m timeout 600 python bench.py --no-cpu-baseline 2>>$O/bench.err | line "phase_multi=$m LSTM" | tee -a $O/ab_phase_multi.txt
done
done
for m in 1 0; do
  MVAE_PHASE_MULTI=$m timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/bench.err | line "phase_multi=$m GRU" | tee -a $O/ab_phase_multi.txt
done
tail -3 $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/bench_lstm_kernel_stats.csv
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 40 > $O/timeline_lstm_step.txt
cat $O/timeline_lstm_step.txt | head -60
