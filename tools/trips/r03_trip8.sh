cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q --timeout 600 --maxfail 10 > $O/pytest_all.txt 2>&1
tail -15 $O/pytest_all.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['value']), round(d['roofline']['us_per_time_step'],3), round(d['roofline']['critical_path']['us_per_step_fwd'],3))"; }
for rep in 1 2; do
for m in 1 0; do
  MVAE_PHASE_MULTI=$m timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/bench.err | line "phase_multi=$m GRU" | tee -a $O/ab_phase_multi.txt
done
done
MVAE_PHASE_MULTI=1 timeout 600 python bench.py --no-cpu-baseline 2>>$O/bench.err | line "phase_multi=1 LSTM" | tee -a $O/ab_phase_multi.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline --cell GRU > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 40 > $O/timeline_gru_step.txt
cat $O/timeline_gru_step.txt | head -70
