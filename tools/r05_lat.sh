cd /root/repo
for c in LSTM GRU; do for z in 64 256; do timeout 120 python tools/latent_bench.py $c $z 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done; done
