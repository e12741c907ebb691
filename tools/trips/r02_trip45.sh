cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 1 0 1 0; do
  MVAE_KSTREAM_DECODER=$v timeout 600 python bench.py --no-cpu-baseline 2>>$O/kd.err | line "kstream_decoder=$v LSTM" >> $O/ab_kd.txt
done
for v in 1 0 1; do
  MVAE_KSTREAM_DECODER=$v timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/kd.err | line "kstream_decoder=$v GRU" >> $O/ab_kd.txt
done
cat $O/ab_kd.txt; tail -3 $O/kd.err
