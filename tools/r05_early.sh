cd /root/repo
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > gpurun_out/pytest_part.txt 2>&1; echo "pytest rc $?"; grep -E "passed|failed|rror" gpurun_out/pytest_part.txt | tail -5
for k in "defer_portions=1" "defer_portions=2" "defer_portions=4"; do
  timeout 300 python tools/knob_bench.py --shape reference --steps 200 $k 2>&1 | tail -1 | cut -c1-170
  timeout 300 python tools/knob_bench.py --shape reference --cell LSTM --steps 200 $k 2>&1 | tail -1 | cut -c1-170
done
