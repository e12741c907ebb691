cd /root/repo
for k in "defer_split_wgs=0" "defer_split_wgs=256" "defer_split_wgs=384" "defer_split_wgs=512" "defer_split_wgs=1024"; do
  timeout 300 python tools/knob_bench.py --shape reference --steps 200 $k 2>&1 | tail -1 | cut -c1-150
  timeout 300 python tools/knob_bench.py --shape reference --cell LSTM --steps 200 $k 2>&1 | tail -1 | cut -c1-150
done
timeout 300 python tools/knob_bench.py --shape bench --cell GRU --steps 60 2>&1 | tail -1 | cut -c1-150
