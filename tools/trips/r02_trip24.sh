cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 0 4 0 4 1 2; do
  MVAE_KSTREAM_GRADS=0 MVAE_EXP_TOUCH_STREAMS=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "kstream=0 touch=$v LSTM" >> $O/ab_touch.txt
done
cat $O/ab_touch.txt
