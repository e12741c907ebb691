cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zz
mkdir -p $O; cd $R
b() { timeout 300 python bench.py --no-cpu-baseline --cell $1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1 $2', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/ab_misc.txt; }
for c in LSTM GRU; do
  b $c "default"
  MVAE_HOLD_DEC_GRADS=2 b $c "hold_dec_grads=2"
  MVAE_HOLD_DEC_GRADS=0 b $c "hold_dec_grads=0"
  MVAE_KSTREAM_WGS=24 b $c "kstream_wgs=24"
  MVAE_KSTREAM_WGS=40 b $c "kstream_wgs=40"
done
