cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches or elbo_trajectory or kstream_gate" > $O/pytest_phase.txt 2>&1
tail -4 $O/pytest_phase.txt
for rep in 1 2; do
for c in LSTM GRU; do for v in 0 1; do
  MVAE_INDEX_DENSE=$v timeout 300 python bench.py --no-cpu-baseline --cell $c 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$c index_dense=$v', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab.txt
done; done; done
