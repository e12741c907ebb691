# round 2, second GPU trip: the whole -m gpu suite with the new staging / DP paths, the e2e fit bench, bench.py, PMC traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
timeout 600 python tools/fit_e2e_bench.py > $O/fit_e2e.txt 2>&1
timeout 600 python tools/fit_e2e_bench.py --with-prepass >> $O/fit_e2e.txt 2>&1
timeout 600 python tools/fit_e2e_bench.py --with-prepass --host-history >> $O/fit_e2e.txt 2>&1
timeout 600 python tools/fit_e2e_bench.py --windows 256 --songs 8 >> $O/fit_e2e.txt 2>&1
cat $O/fit_e2e.txt
timeout 900 python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
tail -3 $O/bench_lstm.err; cat $O/bench_lstm.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py --fetch $(find /tmp/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1) --write $(find /tmp/pmcb_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O/bench_traffic.json
