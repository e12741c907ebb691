cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
b() { timeout 300 python bench.py --no-cpu-baseline --cell $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab8.txt; }
for rep in 1 2; do
  for c in LSTM GRU; do
  b $c chunk16
  MVAE_PIPE_CHUNK=8 b $c chunk8
  MVAE_PIPE_CHUNK=4 b $c chunk4
  done
done
