#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04i
( time python -m pytest tests -x -q -m gpu --durations=12 ) > gpurun_out/r04i/pytest_gpu.txt 2>&1
tail -22 gpurun_out/r04i/pytest_gpu.txt
( time python bench.py ) > gpurun_out/r04i/bench_default.json 2> gpurun_out/r04i/bench_default.err
tail -4 gpurun_out/r04i/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04i/bench_default.json').read().strip().splitlines()[-1])
print("headline: %.1f windows/s %.3f ms/step frac %.4f plan %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v for k,v in d.get("plan",{}).items() if k!="what"}))
print("critical path:", {k:v for k,v in d["roofline"].get("critical_path",{}).items() if k!="what"})
print("elbo", d["elbo"].get("max_abs_diff"), "cpu", d.get("cpu_baseline",{}).get("value"))
for o in d.get("other_configs", []):
    print(" other:", {k: (round(v,3) if isinstance(v,float) else v) for k,v in o.items() if k not in ("workload",)})
PY
