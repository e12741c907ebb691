#!/bin/bash
# round 6: do the schedule knobs tuned for the 4-wave GRU kernels still sit at their optimum with the two-waves-per-SIMD ones?
# each setting in a process of its own (the first engine of a process is the fast one)
out=gpurun_out/knobs; mkdir -p $out
run() { echo "== $*" >> $out/sweep.txt; timeout 300 python tools/knob_bench.py "$@" 2>&1 | tail -2 >> $out/sweep.txt; }
for pc in 8 16 32; do run --shape bench --cell GRU pipe_chunk=$pc; done
for kw in 8 16 24 32; do run --shape bench --cell GRU kstream_wgs=$kw; done
for pc in 8 16 32; do run --shape bench --cell LSTM pipe_chunk=$pc; done
for pc in 8 16; do run --shape reference --cell GRU pipe_chunk=$pc; done
cat $out/sweep.txt
