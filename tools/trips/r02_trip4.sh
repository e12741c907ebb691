# round 2, fourth GPU trip: whole -m gpu suite, e2e fit with host-time breakdown, bench (+cpu baseline), kernel trace + timeline, PMC traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --durations=8 --timeout 400 --maxfail 12 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -25 $O/gpu_tests.log
timeout 600 python tools/fit_e2e_bench.py > $O/fit_e2e.txt 2>&1
timeout 600 python tools/fit_e2e_bench.py --windows 256 --songs 8 >> $O/fit_e2e.txt 2>&1
cat $O/fit_e2e.txt
timeout 900 python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
tail -3 $O/bench_lstm.err; cat $O/bench_lstm.json
timeout 900 python bench.py --cell GRU --no-cpu-baseline > $O/bench_gru.json 2> $O/bench_gru.err
cat $O/bench_gru.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/bench_lstm_kernel_stats.csv
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_lstm_step.txt
head -50 $O/timeline_lstm_step.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --prewarm-max 0 > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py --fetch $(find /tmp/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1) --write $(find /tmp/pmcb_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O/bench_traffic.json
