#!/bin/bash
# round 6: do the two-waves-per-SIMD GRU kernels slow each other down like the 4-wave ones (profiles/r04_q_concurrency.txt)?
# k copies on k streams; pace = slope between T=512 and T=2048 (the fork / join harness drops out)
out=gpurun_out/conc; mkdir -p $out; : > $out/gru_conc.txt
for w8 in "--w8" ""; do
for k in 1 3 6; do
for T in 512 2048; do
  echo "== GRU $w8 concurrent=$k T=$T" >> $out/gru_conc.txt
  timeout 300 python tools/rnn_microbench.py --cell GRU $w8 --concurrent $k --T $T --reps 3 2>&1 | grep -v amdgpu.ids >> $out/gru_conc.txt
done; done; done
echo "== GRU --w8 signal=16 (publishes every 16 steps like a pipelined producer)" >> $out/gru_conc.txt
timeout 300 python tools/rnn_microbench.py --cell GRU --w8 --signal 16 2>&1 | grep -v amdgpu.ids >> $out/gru_conc.txt
cat $out/gru_conc.txt
