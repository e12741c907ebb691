# Round 5: LSTM BPTT, requests ordered by tile pair with a staged wait (BWL_STAGED_WAIT) against round 4's order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
V=$R/build/variants
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
mb() { echo "== $1" >> $O/rnn_microbench.txt; shift; "$@" python tools/rnn_microbench.py --cell LSTM 2>&1 | grep "bwd" >> $O/rnn_microbench.txt; }
mb "staged wait (product)" env
mb "round 4 order" env MVAE_LIB=$V/lib_st0.so
mb "staged wait (product), again" env
mb "round 4 order, again" env MVAE_LIB=$V/lib_st0.so
b() { echo "== $1" >> $O/bench_ab.txt; shift; "$@" python bench.py --no-cpu-baseline --no-other-configs --elbo-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['us_per_time_step'], d['roofline']['launch_ms_by_layer'])" >> $O/bench_ab.txt; }
for rep in 1 2 3; do
b "staged, no companion" env MVAE_L2_TOUCH=0
b "round 4 order, no companion" env MVAE_LIB=$V/lib_st0.so MVAE_L2_TOUCH=0
b "staged + companion" env MVAE_L2_TOUCH=1
b "round 4 order + companion" env MVAE_LIB=$V/lib_st0.so MVAE_L2_TOUCH=1
done
ls -la $O
