cd /root/repo
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "nothing_is_added" 2>&1 | grep -E "passed|failed|Error|assert" | head -8
