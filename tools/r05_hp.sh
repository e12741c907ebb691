cd /root/repo
echo "== GRU fit x5"; for i in 1 2 3 4 5; do timeout 300 python tools/fit_e2e_bench.py --songs 8 --cell GRU 2>&1 | grep -E "^epoch 3|packer threads" | cut -c1-140 | sed 's/end to end.*(/(/' | tr '\n' ' '; echo; done
echo "== LSTM fit x5"; for i in 1 2 3 4 5; do timeout 300 python tools/fit_e2e_bench.py --songs 8 2>&1 | grep -E "^epoch 3" | cut -c1-140 | sed 's/end to end.*(/(/'; done
echo "== LSTM fit with pre-pass x3"; for i in 1 2 3; do timeout 300 python tools/fit_e2e_bench.py --songs 8 --with-prepass 2>&1 | grep -E "^epoch 3" | cut -c1-140 | sed 's/end to end.*(/(/'; done
echo "== training script x3"; for i in 1 2 3; do timeout 300 python tools/training_script_bench.py 2>&1 | grep -E "^epoch 3" | cut -c1-130; done
