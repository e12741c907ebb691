# Round 5: the reference's shipped shape (T=64): deferred weight-gradient GEMMs (one mvae_gemm_multi launch) against launching them beside the recurrences
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -q --deselect "tests/test_baseline_configs_gpu.py" > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
MVAE_PLANS=0 timeout 600 python -m pytest tests/test_dp_fit_gpu.py -m gpu -q -k f32_small > $O/pytest_dp_noplans.txt 2>&1; echo "pytest rc $?" >> $O/pytest_dp_noplans.txt
b() { echo "== $1" >> $O/t64.txt; shift; "$@" python bench.py --config 0 --no-cpu-baseline --steps 100 --warmup 20 --elbo-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['launch_ms_by_layer'], d['plan'])" >> $O/t64.txt; }
for rep in 1 2 3; do
b "deferred weight-gradient GEMMs (default)" env
b "MVAE_DEFER_GRADS_ROWS=0 (round 4 schedule)" env MVAE_DEFER_GRADS_ROWS=0
done
echo "== LSTM, deferred" >> $O/t64.txt; python bench.py --config 0 --cell LSTM --no-cpu-baseline --steps 100 --warmup 20 --elbo-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $O/t64.txt
b "MVAE_DIAG_NO_PARAM_GRADS=1 (timing only: no parameter-gradient launches at all)" env MVAE_DIAG_NO_PARAM_GRADS=1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_c0n -- python bench.py --config 0 --no-cpu-baseline --no-other-configs --steps 30 --warmup 10 --prewarm-max 1 --elbo-steps 0 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_c0n -name "*kernel_trace.csv" | head -1) --min-us 0 > $O/timeline_config0_gru_deferred.txt
ls -la $O
