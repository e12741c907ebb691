#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04e
python -m pytest tests/test_plan_gpu.py tests/test_golden_gpu.py tests/test_dp_gpu.py tests/test_dp_fit_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -15
( time python bench.py ) > gpurun_out/r04e/bench_default.json 2> gpurun_out/r04e/bench_default.err
tail -3 gpurun_out/r04e/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04e/bench_default.json').read().strip().splitlines()[-1])
print("headline: %.1f windows/s %.3f ms/step frac %.4f plan %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("plan")))
print("critical path:", d["roofline"].get("critical_path"))
for o in d.get("other_configs", []):
    print(" other:", {k: (round(v,3) if isinstance(v,float) else v) for k,v in o.items() if k not in ("workload",)})
PY
