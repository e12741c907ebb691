# Round 5: the exact command list behind profiles/r05_p_* (run on the GPU box: gpurun -- 'bash tools/collect_profiles_r05.sh'); STEPS="2 3" re-runs only those
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
want() { [ -z "$STEPS" ] || [[ " $STEPS " == *" $1 "* ]]; }
if want 1; then
# 1. the bench line (CPU baseline + float64 CPU ELBO first, then the GPU phase) with other_configs; GRU; f32 parity mode; one-rank RCCL with and without the early bucket
python bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
python bench.py --cell GRU --no-cpu-baseline --no-other-configs > $O/bench_gru.json 2> $O/bench_gru.err
python bench.py --dtype f32 --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 > $O/bench_lstm_f32.json 2> $O/bench_lstm_f32.err
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/bench_lstm_one_rank_rccl.json
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 1 --no-cpu-baseline --no-other-configs --dp-overlap 1 2>/dev/null | tail -1 > $O/bench_lstm_one_rank_rccl_overlap.json
fi; if want 2; then
# 2. kernel trace + stats of the SAME default command; one replayed step's timeline by queue at configs[1] and at the reference's shipped shape
for c in LSTM GRU; do
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -- python bench.py --no-cpu-baseline --no-other-configs --cell $c > /dev/null 2>&1
  cp $(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1) $O/bench_${c}_kernel_stats.csv
done
for sh in bench reference; do
  rm -rf /tmp/ks_t; timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks_t -- python tools/knob_bench.py --shape $sh --steps 30 > /dev/null 2>&1
  python tools/timeline.py $(find /tmp/ks_t -name "*kernel_trace.csv" | head -1) --min-us $([ $sh = bench ] && echo 20 || echo 0) > $O/timeline_${sh}_replayed_step.txt
done
fi; if want 3; then
# 3. HBM traffic of the dominant kernel: reads (FETCH_SIZE) and writes (TCC_EA0_WRREQ_sum x 64 B) in SEPARATE passes, every pass under timeout
PMC_DEAD=0
pmc() { [ $PMC_DEAD = 1 ] && return 124; timeout -k 5 170 "$@"; rc=$?; [ $rc = 124 ] && PMC_DEAD=1; return $rc; }
for c in FETCH_SIZE TCC_EA0_WRREQ_sum; do
  pmc rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 --prewarm-max 0 > $O/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py --fetch $(find /tmp/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1) --write $(find /tmp/pmcb_TCC_EA0_WRREQ_sum -name "*counter_collection.csv" | head -1) --write-counter TCC_EA0_WRREQ_sum --out $O/bench_traffic.json > $O/pmc_traffic.log 2>&1
fi; if want 4; then
# 4. issue / MFMA counters of the recurrent kernels alone and of the GEMM kernels alone
type pmc > /dev/null 2>&1 || { PMC_DEAD=0; pmc() { [ $PMC_DEAD = 1 ] && return 124; timeout -k 5 170 "$@"; rc=$?; [ $rc = 124 ] && PMC_DEAD=1; return $rc; }; }
PMC_GROUPS=("SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM")
i=0
for g in "${PMC_GROUPS[@]}"; do
  i=$((i+1))
  pmc rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_rnn_$i -- python tools/rnn_microbench.py --cell LSTM > /dev/null 2>&1
  pmc rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_gemm_$i -- python tools/gemm_microbench.py > /dev/null 2>&1
done
python tools/pmc_summary.py $(find /tmp/pmc_rnn_* -name "*counter_collection.csv") > $O/rnn_pmc_summary.txt 2>&1
python tools/pmc_kernels.py --match "gemm|proj_ws" $(find /tmp/pmc_gemm_* -name "*counter_collection.csv") > $O/gemm_pmc_summary.txt 2>&1
fi; if want 5; then
# 5. the tools' own timings
python tools/gemm_microbench.py 2>&1 | grep -v amdgpu > $O/gemm_microbench.txt
python tools/rnn_microbench.py --cell LSTM 2>&1 | grep -v amdgpu > $O/rnn_microbench.txt
python tools/rnn_microbench.py --cell GRU 2>&1 | grep -v amdgpu >> $O/rnn_microbench.txt
for args in "" "--with-prepass" "--windows 256 --songs 8 --with-prepass" "--cell GRU"; do
  echo "== tools/fit_e2e_bench.py $args" >> $O/fit_e2e.txt
  python tools/fit_e2e_bench.py $args 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
done
python tools/training_script_bench.py 2>&1 | grep -v amdgpu > $O/training_script_default.txt
for a in "--shape bench" "--shape reference" "--shape reference --cell LSTM"; do python tools/plan_host_bench.py $a 2>&1 | grep -v amdgpu >> $O/plan_host.txt; done
for args in "--config 2" "--config 5" "--config 5 --cell GRU"; do
  python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
fi; if want 6; then
timeout 1700 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
grep "decisive decode\|config4 decode" $O/pytest_gpu.txt > $O/decode_agreement.txt
fi
ls -la $O
