cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_baseline_configs_gpu.py tests/test_classifier_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 400 --maxfail 8 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -12 $O/gpu_tests.log | cut -c1-250
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export MVAE_NO_WS_GEMM=1; else unset MVAE_NO_WS_GEMM; fi
  timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('no_ws=$v LSTM', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'], d['elbo']['loss_final'])" >> $O/bench.txt
done
unset MVAE_NO_WS_GEMM
timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('GRU', d['ms_per_step'], d['median_ms_per_step'])" >> $O/bench.txt
cat $O/bench.txt
for args in "--config 5" "--config 5 --cell GRU" "--config 5 --cell GRU --gemm-blocks 96" "--config 5 --batch 512" "--config 2" "--config 2 --cell GRU"; do
  timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
MVAE_NO_WS_GEMM=1 timeout 300 python tools/decode_bench.py --config 5 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
cat $O/decode.txt
python tools/large_shape_check.py 2>&1 | grep -v amdgpu | tail -2
