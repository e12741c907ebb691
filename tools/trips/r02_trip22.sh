cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in "1 1 2" "0 1 2" "1 2 2" "1 1 4" "1 1 2" "0 1 2"; do
  set -- $v
  MVAE_GATED_GRADS=$1 MVAE_GATED_STREAMS=$2 MVAE_GATED_LAST=$3 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "gated=$1 streams=$2 last=$3 LSTM" >> $O/ab_gated2.txt
done
cat $O/ab_gated2.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_gated2.txt
