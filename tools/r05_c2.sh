cd /root/repo
for k in "phase_max_B=512" "phase_max_B=512 kstream_max_B=512" "phase_max_B=512 kstream_max_B=512 kstream_wgs=64" "pace_mask=0" "pace_mask=2"; do
  timeout 600 python tools/knob_bench.py --shape config2 --steps 12 $k 2>&1 | tail -1 | cut -c1-150
done
