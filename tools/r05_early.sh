cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|rror" gpurun_out/pytest_gpu.txt | tail -3
timeout 900 python bench.py > gpurun_out/bench_early.json 2> gpurun_out/bench_early.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_early.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['median_ms_per_step'], d['step_ms'], d['plan']['recorded'], d['plan']['replayed'], d['roofline']['frac'], d['elbo'].get('max_abs_diff'))
for o in d.get('other_configs',[]): print(o.get('baseline_config'), o.get('cell'), o.get('ms_per_step'), o.get('error'))
P
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 1 --no-cpu-baseline --no-other-configs --dp-overlap 1 2>/dev/null | tail -1 > gpurun_out/bench_rccl_overlap.json
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_rccl_overlap.json').read().strip().splitlines()[-1])
print("one-rank RCCL overlap:", d['ms_per_step'], d['dp']['plan'], d['dp']['allreduce_ms']['early_decoder_bucket'], d['dp']['allreduce_ms']['late'])
P
