cd /tmp && export TMPDIR=/tmp
cd /root/repo
rm -rf /tmp/ks_t; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_t -- python tools/knob_bench.py --shape bench --cell LSTM --steps 20 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_t -name "*kernel_trace.csv" | head -1) --min-us 15 > gpurun_out/tl_lstm_new.txt
