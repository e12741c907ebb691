cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 400 --maxfail 8 > $O/pytest_full.txt 2>&1
tail -3 $O/pytest_full.txt
for i in $(seq 1 12); do
  timeout 300 python bench.py --no-cpu-baseline 2>>$O/stress.err | python -c "import json,sys; d=json.load(sys.stdin); print('LSTM', round(d['ms_per_step'],3))" >> $O/stress.txt 2>&1
done
for i in $(seq 1 4); do
  timeout 300 python bench.py --no-cpu-baseline --cell GRU 2>>$O/stress.err | python -c "import json,sys; d=json.load(sys.stdin); print('GRU', round(d['ms_per_step'],3))" >> $O/stress.txt 2>&1
done
for i in $(seq 1 12); do
  timeout 300 python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -c "RuntimeError" >> $O/stress_fit.txt
done
tr '\n' ' ' < $O/stress.txt; echo; tr '\n' ' ' < $O/stress_fit.txt; echo
grep -c "Traceback\|Error" $O/stress.err
