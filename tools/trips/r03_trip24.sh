cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03w
mkdir -p $O
cd $R
b() { timeout 300 python bench.py --no-cpu-baseline --cell $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab.txt; }
for rep in 1 2; do
  MVAE_INDEX_DENSE=0 b LSTM "index_dense=0"
  MVAE_INDEX_DENSE=1 b LSTM "index_dense=1"
  MVAE_INDEX_DENSE=0 b GRU "index_dense=0"
  MVAE_INDEX_DENSE=1 b GRU "index_dense=1"
done
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
