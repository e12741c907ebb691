cd /root/repo
for k in 0 64 256 1024; do
  timeout 300 python tools/knob_bench.py --shape reference --steps 200 wait_poll_sleep=$k 2>&1 | tail -1 | cut -c1-170
  timeout 300 python tools/knob_bench.py --shape bench --steps 60 wait_poll_sleep=$k 2>&1 | tail -1 | cut -c1-170
  timeout 300 python tools/knob_bench.py --shape bench --cell GRU --steps 60 wait_poll_sleep=$k 2>&1 | tail -1 | cut -c1-170
done
