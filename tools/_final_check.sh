#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04q
( time python -m pytest tests -x -q -m gpu --durations=8 ) > gpurun_out/r04q/pytest_gpu.txt 2>&1
tail -16 gpurun_out/r04q/pytest_gpu.txt
( time python bench.py ) > gpurun_out/r04q/bench_default.json 2> gpurun_out/r04q/bench_default.err
tail -3 gpurun_out/r04q/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04q/bench_default.json').read().strip().splitlines()[-1])
print("headline: %.1f windows/s %.3f ms/step frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for o in d.get("other_configs", []):
    print(" other: cfg %s %s %.3f ms %.0f w/s frac %.4f" % (o.get("baseline_config"), o.get("cell"), o.get("ms_per_step", 0), o.get("value", 0), o.get("roofline", {}).get("frac", 0)))
PY
