cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline 2>>$O/j1.err | line "one end gate LSTM" >> $O/ab_j1.txt; done
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/j1.err | line "one end gate GRU" >> $O/ab_j1.txt; done
cat $O/ab_j1.txt
