cd /tmp && export TMPDIR=/tmp
cd /root/repo
rm -rf /tmp/ks_t; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_t -- python tools/knob_bench.py --shape bench --cell GRU --steps 20 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_t -name "*kernel_trace.csv" | head -1) --min-us 15 > gpurun_out/tl_gru_k14.txt
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/pytest_part.txt 2>&1; echo "pytest rc $?"; grep -E "passed|failed|rror" gpurun_out/pytest_part.txt | tail -3
