cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
python bench.py --cell GRU --no-cpu-baseline > $O/bench_gru.json 2> $O/bench_gru.err
for i in 1 2 3 4 5 6; do timeout 300 python bench.py --no-cpu-baseline --cell GRU 2>>$O/s.err | python -c "import json,sys; d=json.load(sys.stdin); print('GRU', round(d['ms_per_step'],3))" >> $O/stress_gru.txt 2>&1; done
for i in 1 2 3 4; do timeout 300 python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -c RuntimeError >> $O/stress_gru.txt; done
tr '\n' ' ' < $O/stress_gru.txt; echo
timeout 2400 python -m pytest tests -m gpu -q --timeout 400 --maxfail 8 > $O/pytest_full2.txt 2>&1
tail -3 $O/pytest_full2.txt
python bench.py > $O/bench_lstm_default.json 2> $O/bench_lstm_default.err
tail -c 300 $O/bench_lstm_default.json
