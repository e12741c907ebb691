cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches" > $O/pytest_phase2.txt 2>&1
tail -4 $O/pytest_phase2.txt
for rep in 1 2; do
for c in LSTM GRU; do
  MVAE_INDEX_DENSE=0 timeout 300 python bench.py --no-cpu-baseline --cell $c 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$c index_dense=0', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab3.txt
for v in 16 32 64; do
  MVAE_INDEX_DENSE_BLOCKS=$v timeout 300 python bench.py --no-cpu-baseline --cell $c 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$c index_dense blocks=$v', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab3.txt
done; done; done
