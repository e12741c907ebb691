cd /root/repo
for th in 8 16 32 64 128; do
echo "== threads $th"
for i in 1 2 3; do
timeout 300 python tools/fit_e2e_bench.py --songs 8 --threads $th 2>&1 | grep -A1 -E "^epoch 3" | cut -c1-200 | sed 's/end to end.*(/(/' | tr '\n' ' ' | sed 's/ | pre-pass.*host time inside fit per 256-window step:/ |/'; echo
done
done
