cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
for rep in 1 2; do
for c in LSTM GRU; do for v in 8 16 32; do
  MVAE_INDEX_DENSE_BLOCKS=$v timeout 300 python bench.py --no-cpu-baseline --cell $c 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$c index_dense blocks=$v', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab2.txt
done; done; done
