cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x > $O/pytest_subset.txt 2>&1
tail -6 $O/pytest_subset.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['value']), round(d['roofline']['us_per_time_step'],3), round(d['roofline']['critical_path']['us_per_step_fwd'],3))"; }
for rep in 1 2; do
for c in LSTM GRU; do
for m in 1 0; do
  MVAE_PHASE_MULTI=$m timeout 600 python bench.py --no-cpu-baseline --cell $c 2>>$O/bench.err | line "phase_multi=$m $c" | tee -a $O/ab_phase_multi.txt
done
done
done
for c in LSTM GRU; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -- python $R/bench.py --no-cpu-baseline --cell $c > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/ks_$c -name "*kernel_trace.csv" | head -1) --min-us 60 > $O/timeline_${c}_step.txt
done
cp $(find /tmp/ks_LSTM -name "*kernel_stats.csv" | head -1) $O/bench_lstm_kernel_stats.csv
cat $O/timeline_LSTM_step.txt | head -50
grep -v "us  +" $O/timeline_GRU_step.txt; grep "multi_k\|latent_chain\|head_k<unsigned short, 4\|adam\|prepare" $O/timeline_GRU_step.txt
