cd /root/repo
O=gpurun_out/elbo; mkdir -p $O
for c in LSTM GRU; do
  timeout 1500 python tests/studies/elbo_long.py --cell $c --lr 2e-4 > $O/elbo_${c}.txt 2>&1
  tail -3 $O/elbo_${c}.txt
done
