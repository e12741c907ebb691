cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
for c in LSTM GRU; do timeout 300 python bench.py --config 4 --no-cpu-baseline --steps 20 --warmup 5 --cell $c > $O/bench_config4_$(echo $c | tr A-Z a-z).json 2>/dev/null; done
python -c "
import json
for c in ('lstm','gru'):
    d=json.load(open('$O/bench_config4_%s.json'%c)); print(c, round(d['ms_per_step'],3), round(d['value']))"
