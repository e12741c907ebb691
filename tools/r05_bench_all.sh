cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
python bench.py --cell GRU --no-cpu-baseline --no-other-configs > $O/bench_gru.json 2> $O/bench_gru.err
python bench.py --no-cpu-baseline --no-other-configs > $O/bench_lstm2.json 2> $O/bench_lstm2.err
tail -3 $O/pytest_gpu.txt; tail -3 $O/bench_lstm.err
python - <<'PY'
import json
for f in ("bench_lstm.json","bench_gru.json","bench_lstm2.json"):
    d=json.loads(open("/root/repo/gpurun_out/r05r/"+f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, d["ms_per_step"], d["value"], r["frac"], r["avg_launch_ms"], r["launches"], d["plan"])
    for o in d.get("other_configs", []):
        print("   ", str(o.get("workload"))[:80], o.get("ms_per_step"), o.get("value"))
PY
