cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 8192 2048 4096 8192 2048 1024; do
  MVAE_SPLIT_ROWS=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "split_rows=$v LSTM" >> $O/ab_split.txt
done
for v in 8192 2048; do
  MVAE_SPLIT_ROWS=$v timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "split_rows=$v GRU" >> $O/ab_split.txt
done
cat $O/ab_split.txt
