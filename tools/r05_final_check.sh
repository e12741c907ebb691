cd /root/repo
O=gpurun_out/r05f3; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.txt
grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
timeout 900 python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
for args in "" "--with-prepass" "--windows 256 --songs 8 --with-prepass" "--cell GRU"; do
  echo "== tools/fit_e2e_bench.py $args" >> $O/fit_e2e.txt
  python tools/fit_e2e_bench.py $args 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
done
python tools/training_script_bench.py 2>&1 | grep -v amdgpu > $O/training_script_default.txt
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05f3/bench_lstm.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['median_ms_per_step'], d['step_ms'], d['roofline']['frac'], d['elbo'].get('max_abs_diff'), d['cpu_baseline']['value'])
for o in d.get('other_configs',[]): print(o.get('baseline_config'), o.get('cell'), o.get('ms_per_step'), o.get('value'), o.get('error'))
P
grep "^epoch 3" $O/fit_e2e.txt | cut -c1-140; grep "^epoch [23]" $O/training_script_default.txt | cut -c1-150
