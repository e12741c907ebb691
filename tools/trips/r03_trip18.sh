cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches" > $O/pytest_phase3.txt 2>&1
tail -4 $O/pytest_phase3.txt
b() { timeout 300 python bench.py --no-cpu-baseline --cell $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab4.txt; }
for rep in 1 2; do
for c in LSTM GRU; do
  MVAE_INDEX_DENSE=0 b $c "index_dense=0 xpand=64"
  MVAE_INDEX_DENSE=0 MVAE_XPAND_BLOCKS=16 b $c "index_dense=0 xpand=16"
  MVAE_INDEX_DENSE_BLOCKS=8 MVAE_XPAND_BLOCKS=16 b $c "index_dense=8 xpand=16"
  MVAE_INDEX_DENSE_BLOCKS=16 MVAE_XPAND_BLOCKS=16 b $c "index_dense=16 xpand=16"
  MVAE_INDEX_DENSE_BLOCKS=32 MVAE_XPAND_BLOCKS=16 b $c "index_dense=32 xpand=16"
  MVAE_INDEX_DENSE_BLOCKS=32 MVAE_XPAND_BLOCKS=64 b $c "index_dense=32 xpand=64"
done; done
