cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zz
mkdir -p $O; cd $R
MVAE_TRAIN_HEAD_SLICES=4 timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches or elbo_trajectory" > $O/pytest_ths.txt 2>&1
tail -3 $O/pytest_ths.txt
b() { timeout 300 python bench.py --no-cpu-baseline --cell $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/ab_ths.txt; }
for c in LSTM GRU; do
  MVAE_TRAIN_HEAD_SLICES=1 b $c "head slices 1"
  MVAE_TRAIN_HEAD_SLICES=2 b $c "head slices 2"
  MVAE_TRAIN_HEAD_SLICES=4 b $c "head slices 4"
done
