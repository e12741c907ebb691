cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_L -- python bench.py --no-cpu-baseline --cell LSTM > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_L -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_LSTM_step.txt
head -70 $O/timeline_LSTM_step.txt | cut -c1-120
