cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in none instr grad2 instr,grad2 none instr,grad2; do
  MVAE_KSTREAM_GRADS=0 MVAE_ALIAS_STREAMS=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "alias=$v LSTM" >> $O/ab_alias.txt
done
for q in 8 10; do
  GPU_MAX_HW_QUEUES=$q MVAE_KSTREAM_GRADS=0 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "hwq=$q LSTM" >> $O/ab_alias.txt
done
cat $O/ab_alias.txt
