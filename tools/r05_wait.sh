cd /root/repo
for k in "pace_mask=14" "pace_mask=6"; do
  timeout 300 python tools/knob_bench.py --shape bench --steps 60 $k 2>&1 | tail -1 | cut -c1-150
  timeout 300 python tools/knob_bench.py --shape bench --cell GRU --steps 60 $k 2>&1 | tail -1 | cut -c1-150
  timeout 300 python tools/knob_bench.py --shape reference --steps 200 $k 2>&1 | tail -1 | cut -c1-150
done
