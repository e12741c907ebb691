cd /root/repo
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_dp_fit_gpu.py tests/test_golden_gpu.py tests/test_plan_gpu.py -x -q -m gpu > gpurun_out/pytest_part.txt 2>&1; echo "pytest rc $?"; grep -E "passed|failed|rror" gpurun_out/pytest_part.txt | tail -5
for i in 1 2; do timeout 300 python tools/training_script_bench.py 2>&1 | grep -A1 -E "^epoch [23]" | cut -c1-200; done
