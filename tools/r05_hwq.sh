cd /root/repo
for q in 16 8; do
for n in 0 2 4 6 8 10 12 14 16 20 24; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/knob_bench.py --shape reference --steps 150 --skip-streams $n 2>&1 | tail -1 | cut -c1-150 | sed "s/^/q=$q /"
done
done
