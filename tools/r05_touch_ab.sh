# Round 5: in-kernel L2 touch of the LSTM BPTT (BWL_TOUCH_AHEAD) against the round-4 companion kernel and against neither
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
V=$R/build/variants
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
mb() { echo "== $1" >> $O/rnn_microbench.txt; shift; "$@" python tools/rnn_microbench.py --cell LSTM 2>&1 | grep "bwd" >> $O/rnn_microbench.txt; }
mb "product (TA=2)" env
mb "TA=0 (round 4 kernel)" env MVAE_LIB=$V/lib_ta0.so
mb "dry (TA=2 without the touch instructions)" env MVAE_LIB=$V/lib_dry.so
mb "TA=1" env MVAE_LIB=$V/lib_ta1.so
mb "TA=4" env MVAE_LIB=$V/lib_ta4.so
echo "== TA=0 + companion (--signal 16 --prefetch 8)" >> $O/rnn_microbench.txt
MVAE_LIB=$V/lib_ta0.so python tools/rnn_microbench.py --cell LSTM --signal 16 --prefetch 8 2>&1 | grep bwd >> $O/rnn_microbench.txt
b() { echo "== $1" >> $O/bench_ab.txt; shift; "$@" python bench.py --no-cpu-baseline --no-other-configs --elbo-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['us_per_time_step'], d['roofline']['launch_ms_by_layer'])" >> $O/bench_ab.txt; }
for rep in 1 2; do
b "in-kernel touch only (product lib, MVAE_L2_TOUCH=0)" env MVAE_L2_TOUCH=0
b "round 4: TA=0 + companion" env MVAE_LIB=$V/lib_ta0.so MVAE_L2_TOUCH=1
b "neither: TA=0, MVAE_L2_TOUCH=0" env MVAE_LIB=$V/lib_ta0.so MVAE_L2_TOUCH=0
b "dry, no companion" env MVAE_LIB=$V/lib_dry.so MVAE_L2_TOUCH=0
done
ls -la $O
