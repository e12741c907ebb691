cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
MVAE_STEPS_IN_FLIGHT=0 python bench.py --no-cpu-baseline > $O/bench_lstm_unpaced.json 2> $O/bench_lstm_unpaced.err
python bench.py --cell GRU --no-cpu-baseline --no-other-configs > $O/bench_gru.json 2> $O/bench_gru.err
tail -3 $O/pytest_gpu.txt
python - <<'PY'
import json
for f in ("bench_lstm.json","bench_lstm_unpaced.json","bench_gru.json"):
    d=json.loads(open("/root/repo/gpurun_out/r05n/"+f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["plan"]["host_ms_per_step"])
    for o in d.get("other_configs", []):
        print("   ", o.get("workload", o.get("config"))[:70] if isinstance(o.get("workload", ""), str) else "", o.get("ms_per_step"), o.get("value"))
PY
