cd /root/repo
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_dp_fit_gpu.py -x -q -m gpu > gpurun_out/pytest_part.txt 2>&1; echo "pytest rc $?"; grep -E "passed|failed|rror" gpurun_out/pytest_part.txt | tail -5
timeout 200 python tools/dp_knob_bench.py --overlap 0 2>&1 | grep "one-rank"
for k in "" "hold_side_heads=0" "_hold_dec_grads=0 hold_side_heads=0"; do
  timeout 200 python tools/dp_knob_bench.py --overlap 1 $k 2>&1 | grep "one-rank"
done
timeout 200 python tools/dp_knob_bench.py --overlap 0 --cell GRU 2>&1 | grep "one-rank"
timeout 200 python tools/dp_knob_bench.py --overlap 1 --cell GRU 2>&1 | grep "one-rank"
