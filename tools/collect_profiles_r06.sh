# Round 6: the command list behind profiles/r06_p_* (run on the GPU box: gpurun -- 'bash tools/collect_profiles_r06.sh'); STEPS="2 3" re-runs only those (the default is steps 1-3: STEPS="1 2 3"; 4 and 5 are the sweeps).
# (Issue / MFMA counters of the GRU kernels: tools/collect_pmc_gru_r06.sh -> r06_f_*; HBM traffic of the LSTM kernel: unchanged since round 5.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
STEPS=${STEPS:-"1 2 3"}
want() { [[ " $STEPS " == *" $1 "* ]]; }
if want 1; then
# 1. the bench line (CPU baseline + float64 CPU ELBO first, then the GPU phase, other_configs, fit_e2e); GRU with both kernel families; one-rank RCCL group (policy probe)
python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
python bench.py --cell GRU --no-cpu-baseline --no-other-configs > $O/bench_gru.json 2> $O/bench_gru.err
MVAE_GRU_W8=0 python bench.py --cell GRU --no-cpu-baseline --no-other-configs > $O/bench_gru_4wave.json 2> $O/bench_gru_4wave.err
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_lstm_one_rank_rccl.json
fi; if want 2; then
# 2. kernel trace + stats of the SAME default command, both cells; one replayed GRU step's timeline by queue
for c in LSTM GRU; do
  rm -rf /tmp/ks_$c; timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -- python bench.py --no-cpu-baseline --no-other-configs --cell $c > /dev/null 2>&1
  cp $(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1) $O/bench_${c}_kernel_stats.csv
done
rm -rf /tmp/ks_t; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_t -- python tools/knob_bench.py --shape bench --cell GRU --steps 30 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_t -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_bench_gru_replayed_step.txt
fi; if want 3; then
# 3. the tools' own timings
python tools/rnn_microbench.py --cell LSTM 2>&1 | grep -v amdgpu > $O/rnn_microbench.txt
echo "== GRU, two waves per SIMD (the product)" >> $O/rnn_microbench.txt
python tools/rnn_microbench.py --cell GRU --w8 2>&1 | grep -v amdgpu >> $O/rnn_microbench.txt
echo "== GRU, one wave per SIMD (MVAE_GRU_W8=0)" >> $O/rnn_microbench.txt
python tools/rnn_microbench.py --cell GRU 2>&1 | grep -v amdgpu >> $O/rnn_microbench.txt
for args in "" "--with-prepass" "--cell GRU"; do
  echo "== tools/fit_e2e_bench.py $args" >> $O/fit_e2e.txt
  python tools/fit_e2e_bench.py $args 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
done
python tools/training_script_bench.py 2>&1 | grep -v amdgpu > $O/training_script_default.txt
fi; if want 4; then
# 4. schedule knobs re-swept with the two-waves-per-SIMD GRU kernels (each setting in a process of its own: the first engine of a process is the fast one) -> r06_h_knob_sweep.txt, r06_k_kstream_rows.txt
run() { echo "== $*" >> $O/knob_sweep.txt; timeout 300 python tools/knob_bench.py "$@" 2>&1 | tail -1 >> $O/knob_sweep.txt; }
for pc in 8 16 32; do run --shape bench --cell GRU pipe_chunk=$pc; done
for kw in 8 16 24 32; do run --shape bench --cell GRU kstream_wgs=$kw; done
for kr in 0 1024 2048; do run --shape bench --cell GRU kstream_rows=$kr; done
for pc in 8 16 32; do run --shape bench --cell LSTM pipe_chunk=$pc; done
for kr in 0 2048; do run --shape bench --cell LSTM kstream_rows=$kr; done
for pc in 8 16; do run --shape reference --cell GRU pipe_chunk=$pc; done
fi; if want 5; then
# 5. do the two-waves-per-SIMD GRU kernels slow each other down like the 4-wave ones (profiles/r04_q_concurrency.txt)?  k copies on k streams;
#    the pace is the slope between T=512 and T=2048 (the fork / join harness drops out) -> r06_h_gru_concurrency.txt
for w8 in "--w8" ""; do for k in 1 3 6; do for T in 512 2048; do
  echo "== GRU $w8 concurrent=$k T=$T" >> $O/gru_concurrency.txt
  timeout 300 python tools/rnn_microbench.py --cell GRU $w8 --concurrent $k --T $T --reps 3 2>&1 | grep -v amdgpu.ids >> $O/gru_concurrency.txt
done; done; done
fi
ls -la $O
