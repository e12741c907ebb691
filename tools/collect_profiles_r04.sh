# Round 4: the exact command list behind profiles/r04_p_* (run on the GPU box: gpurun -- 'bash tools/collect_profiles_r04.sh')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
# STEPS="4 5" re-runs only those steps
want() { [ -z "$STEPS" ] || [[ " $STEPS " == *" $1 "* ]]; }
if want 1; then
# 1. the bench line (CPU baseline + float64 CPU ELBO first, then the GPU phase), LSTM and GRU; the other BASELINE configs; f32 parity mode
python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
python bench.py --cell GRU --no-cpu-baseline --no-other-configs > $O/bench_gru.json 2> $O/bench_gru.err
python bench.py --config 2 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_config2_lstm.json 2> $O/bench_cfg.err
python bench.py --config 2 --no-cpu-baseline --steps 10 --warmup 3 --cell GRU > $O/bench_config2_gru.json 2>> $O/bench_cfg.err
python bench.py --config 4 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_config4_lstm.json 2>> $O/bench_cfg.err
python bench.py --config 4 --no-cpu-baseline --steps 20 --warmup 5 --cell GRU > $O/bench_config4_gru.json 2>> $O/bench_cfg.err
python bench.py --config 0 --no-cpu-baseline --steps 100 --warmup 20 > $O/bench_config0_gru.json 2>> $O/bench_cfg.err
python bench.py --config 0 --cell LSTM --no-cpu-baseline --steps 100 --warmup 20 > $O/bench_config0_lstm.json 2>> $O/bench_cfg.err
python bench.py --dtype f32 --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $O/bench_lstm_f32.json 2> $O/bench_lstm_f32.err
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_lstm_one_rank_rccl.json
# ... and with the overlap schedule that bench.py uses for --gpus > 1 (decoder-side bucket reduced beside the encoder BPTT)
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 1 --no-cpu-baseline --no-other-configs --dp-overlap 1 2>/dev/null | tail -1 > $O/bench_lstm_one_rank_rccl_overlap.json
fi; if want 2; then
# 2. kernel trace + stats of the SAME default command, and one step's timeline by queue (LSTM and GRU)
for c in LSTM GRU; do
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -- python bench.py --no-cpu-baseline --no-other-configs --cell $c > /dev/null 2>&1
  cp $(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1) $O/bench_${c}_kernel_stats.csv
  python tools/timeline.py $(find /tmp/ks_$c -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_${c}_step.txt
done
# ... and of one step at configs[2]'s shape (T=2048, 512 windows: gradient time portions)
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_c2 -- python bench.py --config 2 --no-cpu-baseline --no-other-configs --steps 4 --warmup 2 --prewarm-max 1 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/ks_c2 -name "*kernel_trace.csv" | head -1) --min-us 100 > $O/timeline_config2_LSTM_step.txt
fi; if want 3; then
# 3. HBM traffic of the dominant kernel: reads (FETCH_SIZE) and writes in SEPARATE passes.  Writes as TCC_EA0_WRREQ_sum x 64 B:
#    a `--pmc WRITE_SIZE` pass hangs in rocprofv3's start-up on this image (it cost a whole gpurun limit once); the two were
#    calibrated equal in round 2.  Every profiler pass under `timeout -k`: a hung one must not eat the passes behind it.
# (a counter pass that hangs in the profiler's start-up is killed by its timeout; the FIRST one that does ends counter collection
#  for this run - on 2026-09-30 every pass after the first hung on one box and the script spent 45 minutes in time-outs)
PMC_DEAD=0
pmc() { [ $PMC_DEAD = 1 ] && return 124; timeout -k 5 170 "$@"; rc=$?; [ $rc = 124 ] && PMC_DEAD=1; return $rc; }
for c in FETCH_SIZE TCC_EA0_WRREQ_sum; do
  pmc rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 --prewarm-max 0 > $O/pmc_$c.log 2>&1
  grep "bwd_il_k" $(find /tmp/pmcb_$c -name "*counter_collection.csv" | head -1) | cut -c1-400 > $O/pmc_${c}_bwd_rows.csv
done
python tools/pmc_traffic.py --fetch $(find /tmp/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1) --write $(find /tmp/pmcb_TCC_EA0_WRREQ_sum -name "*counter_collection.csv" | head -1) --write-counter TCC_EA0_WRREQ_sum --out $O/bench_traffic.json > $O/pmc_traffic.log 2>&1
fi; if want 4; then
# 4. issue / MFMA counters (NOT `GROUPS`: that name is bash's own array of group ids and cannot be assigned): the recurrent kernels alone, the GEMM kernels alone, decoder inference (configs[4] share)
type pmc > /dev/null 2>&1 || { PMC_DEAD=0; pmc() { [ $PMC_DEAD = 1 ] && return 124; timeout -k 5 170 "$@"; rc=$?; [ $rc = 124 ] && PMC_DEAD=1; return $rc; }; }
PMC_GROUPS=("SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum")
i=0
for g in "${PMC_GROUPS[@]}"; do
  i=$((i+1))
  pmc rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_rnn_$i -- python tools/rnn_microbench.py --cell LSTM > /dev/null 2>&1
  pmc rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_gemm_$i -- python tools/gemm_microbench.py > /dev/null 2>&1
  pmc rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_dec_$i -- python tools/decode_bench.py --config 5 --reps 2 > /dev/null 2>&1
done
python tools/pmc_summary.py $(find /tmp/pmc_rnn_* -name "*counter_collection.csv") > $O/rnn_pmc_summary.txt 2>&1
python tools/pmc_kernels.py --match "gemm|proj_ws" $(find /tmp/pmc_gemm_* -name "*counter_collection.csv") > $O/gemm_pmc_summary.txt 2>&1
python tools/pmc_kernels.py --match "proj_ws|fwd_il_k|fwd_multi|head_k" $(find /tmp/pmc_dec_* -name "*counter_collection.csv") > $O/decode_pmc_summary.txt 2>&1
fi; if want 5; then
# 5. the tools' own timings
python tools/gemm_microbench.py 2>&1 | grep -v amdgpu > $O/gemm_microbench.txt
python tools/rnn_microbench.py --cell LSTM 2>&1 | grep -v amdgpu > $O/rnn_microbench.txt
python tools/rnn_microbench.py --cell GRU 2>&1 | grep -v amdgpu >> $O/rnn_microbench.txt
# ... the LSTM / GRU BPTT kernel with its L2-touch companion beside it (mvae_l2_touch_bwd following the published chunks)
for c in LSTM GRU; do echo "== $c --signal 16 --prefetch 8 (T=512, then T=2048)" >> $O/rnn_microbench_l2_touch.txt; for t in 512 2048; do python tools/rnn_microbench.py --cell $c --signal 16 --prefetch 8 --T $t --reps 8 2>&1 | grep "bwd" >> $O/rnn_microbench_l2_touch.txt; done; done
# ... the generic kernels of the f32 parity mode (weight fragments streamed through a ring of in-flight loads)
for c in LSTM GRU; do echo "== $c --f32" >> $O/rnn_microbench_f32.txt; python tools/rnn_microbench.py --cell $c --f32 --reps 3 2>&1 | grep -v amdgpu >> $O/rnn_microbench_f32.txt; done
for args in "" "--with-prepass" "--windows 256 --songs 8" "--windows 256 --songs 8 --with-prepass" "--with-prepass --lazy"; do
  echo "== tools/fit_e2e_bench.py $args" >> $O/fit_e2e.txt
  python tools/fit_e2e_bench.py $args 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
done
for args in "--config 2" "--config 5" "--config 5 --cell GRU"; do
  python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
python tools/decode_product_bench.py 2>&1 | grep -v amdgpu >> $O/decode.txt
python tools/large_shape_check.py 2>&1 | grep -v amdgpu > $O/large_shape.txt
# host enqueue against device time, Python enqueue vs step-plan replay: BASELINE configs[1] and the reference's shipped configuration
for a in "--shape bench" "--shape reference" "--shape reference --cell LSTM"; do python tools/plan_host_bench.py $a 2>&1 | grep -v amdgpu >> $O/reference_default.txt; done
fi
ls -la $O
