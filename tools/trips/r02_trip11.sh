cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
for v in 2 1 4 2 1 4; do
  MVAE_ONEHOT_SPLIT=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('onehot_split=$v', d['ms_per_step'], d['median_ms_per_step'], d['elbo']['loss_final'])" >> $O/ab_onehot.txt
done
cat $O/ab_onehot.txt
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 400 -x 2>&1 | tail -3
