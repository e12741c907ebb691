cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
python tools/fit_e2e_bench.py 2>&1 | grep -v amdgpu > $O/fit_e2e.txt
python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
python tools/fit_e2e_bench.py --windows 256 --songs 8 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
grep -c Traceback $O/fit_e2e.txt
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --no-cpu-baseline 2>>$O/stress.err | python -c "import json,sys; d=json.load(sys.stdin); print('stress LSTM', d['ms_per_step'])" >> $O/stress.txt 2>&1
  timeout 300 python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -c "Traceback\|timed out" >> $O/stress.txt
done
timeout 300 python bench.py --no-cpu-baseline --cell GRU 2>>$O/stress.err | python -c "import json,sys; d=json.load(sys.stdin); print('stress GRU', d['ms_per_step'])" >> $O/stress.txt 2>&1
cat $O/stress.txt
