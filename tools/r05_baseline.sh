# Round 5, first call: this box's baseline before any change (bench line, microbench, T=64 timeline of the reference's shipped shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline --no-other-configs > $O/bench_lstm.json 2> $O/bench_lstm.err
python bench.py --cell GRU --no-cpu-baseline --no-other-configs > $O/bench_gru.json 2> $O/bench_gru.err
python bench.py --config 0 --no-cpu-baseline --steps 100 --warmup 20 > $O/bench_config0_gru.json 2>> $O/bench_cfg.err
python tools/rnn_microbench.py --cell LSTM 2>&1 | grep -v amdgpu > $O/rnn_microbench.txt
python tools/rnn_microbench.py --cell GRU 2>&1 | grep -v amdgpu >> $O/rnn_microbench.txt
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_c0 -- python bench.py --config 0 --no-cpu-baseline --no-other-configs --steps 30 --warmup 10 --prewarm-max 1 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_c0 -name "*kernel_trace.csv" | head -1) --min-us 0 > $O/timeline_config0_gru.txt
python tools/plan_host_bench.py --shape reference 2>&1 | grep -v amdgpu > $O/plan_host_reference.txt
ls -la $O
