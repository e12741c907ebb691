cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_baseline_configs_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 400 --maxfail 12 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log | cut -c1-250
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('LSTM', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])" >> $O/bench.txt
done
timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('GRU', d['ms_per_step'], d['median_ms_per_step'])" >> $O/bench.txt
cat $O/bench.txt
for args in "--config 5" "--config 5 --gemm-blocks 96" "--config 5 --cell GRU" "--config 5 --batch 256" "--config 2"; do
  timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
cat $O/decode.txt
python tools/large_shape_check.py LSTM 2>&1 | grep -v amdgpu | tail -2
