cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
MVAE_VALUE_JOIN=1 timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches or elbo_trajectory or alternating" > $O/pytest_vj.txt 2>&1
tail -3 $O/pytest_vj.txt
b() { timeout 300 python bench.py --no-cpu-baseline --cell $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/ab_vj.txt; }
for rep in 1 2; do
  for c in LSTM GRU; do
    MVAE_VALUE_JOIN=0 b $c "event joins"
    MVAE_VALUE_JOIN=1 b $c "value joins"
  done
done
