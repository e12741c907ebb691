cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3))"; }
for v in 32 24 48 32 24; do
  MVAE_KSTREAM_WGS=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line "LSTM wgs=$v" >> $O/ab_lstm_wgs.txt
done
cat $O/ab_lstm_wgs.txt
