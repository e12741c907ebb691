cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
# new tests first (fast feedback), then everything
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "fused_history or deferred_latent or chip_filling or alternating or kstream_gate or elbo_trajectory" -s > $O/pytest_new.txt 2>&1
tail -30 $O/pytest_new.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout 600 --maxfail 10 > $O/pytest_all.txt 2>&1
tail -15 $O/pytest_all.txt
for n in 256 1024; do
  timeout 600 python tools/fit_e2e_bench.py --songs 4 --windows $n > $O/fit_e2e_$n.txt 2>&1
  timeout 600 python tools/fit_e2e_bench.py --songs 4 --windows $n --with-prepass > $O/fit_e2e_${n}_prepass.txt 2>&1
  tail -3 $O/fit_e2e_$n.txt $O/fit_e2e_${n}_prepass.txt
done
timeout 600 python bench.py --no-cpu-baseline > $O/bench_lstm.json 2> $O/bench_lstm.err
python -c "import json; d=json.load(open('$O/bench_lstm.json')); print('LSTM', d['ms_per_step'], d['median_ms_per_step'])"
