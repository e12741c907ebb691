cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
b() { timeout 300 python bench.py --no-cpu-baseline --cell $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench_ab7.txt; }
for rep in 1 2; do
  b LSTM product
  for v in c56l2 c58l2 c56l4 c58; do MVAE_LIB=$R/build/variants/lib_$v.so b LSTM $v; done
done
