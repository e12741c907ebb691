cd /root/repo
for r in 1 2; do
MVAE_BENCH_STEP_TIMES=1 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_new$r.json 2> gpurun_out/bench_new$r.err
grep "headline ms per step" gpurun_out/bench_new$r.err | cut -c1-200
python - <<P
import json
d=json.loads(open('gpurun_out/bench_new$r.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['median_ms_per_step'], d['step_ms'], d['plan']['recorded'], d['plan']['replayed'], d['roofline']['frac'])
for o in d.get('other_configs',[]): print(o.get('baseline_config'), o.get('cell'), o.get('ms_per_step'), o.get('error'))
P
done
