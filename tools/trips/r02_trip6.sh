cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_classifier_gpu.py -m gpu -q --timeout 400 --maxfail 12 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -40 $O/gpu_tests.log | cut -c1-250
for v in 1 0 1 0; do
  MVAE_TAIL_ON_MAIN=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('tail_on_main=$v', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['traffic'])" >> $O/ab_tail.txt
done
cat $O/ab_tail.txt
timeout 600 python tools/fit_e2e_bench.py > $O/fit_e2e.txt 2>&1
timeout 600 python tools/fit_e2e_bench.py --windows 256 --songs 8 >> $O/fit_e2e.txt 2>&1
grep -v amdgpu $O/fit_e2e.txt
for args in "--config 5" "--config 5 --gemm-blocks 32" "--config 5 --gemm-blocks 48" "--config 5 --split 2" "--config 5 --split 4" "--config 5 --batch 512" "--config 5 --batch 256" "--config 2"; do
  timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
cat $O/decode.txt
