cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|rror" gpurun_out/pytest_gpu.txt | tail -3
timeout 900 python bench.py > gpurun_out/bench_early.json 2> gpurun_out/bench_early.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_early.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['median_ms_per_step'], d['step_ms'], d['plan']['recorded'], d['plan']['replayed'], d['roofline']['frac'])
for o in d.get('other_configs',[]): print(o.get('baseline_config'), o.get('cell'), o.get('ms_per_step'), o.get('error'))
P
