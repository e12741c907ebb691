# Round 2: the exact command list behind profiles/r02_p_* (run on the GPU box: gpurun -- 'bash tools/collect_profiles_r02.sh')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02p
mkdir -p $O
# 1. the bench line (CPU baseline first, then the GPU phase), LSTM and GRU; f32 "parity mode" beside them
python $R/bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
python $R/bench.py --cell GRU --no-cpu-baseline > $O/bench_gru.json 2> $O/bench_gru.err
python $R/bench.py --dtype f32 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_lstm_f32.json 2> $O/bench_lstm_f32.err
# 2. kernel trace + stats of the SAME default command, and one step's timeline by queue
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/bench_lstm_kernel_stats.csv
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 20 > $O/timeline_lstm_step.txt
# 3. HBM traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 counter slots: 3 + 2 do not fit)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --prewarm-max 0 > $O/pmc_$c.log 2>&1
  grep "bwd_il_k" $(find /tmp/pmcb_$c -name "*counter_collection.csv" | head -1) | cut -c1-400 > $O/pmc_${c}_bwd_rows.csv
done
python $R/tools/pmc_traffic.py --fetch $(find /tmp/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1) --write $(find /tmp/pmcb_WRITE_SIZE -name "*counter_collection.csv" | head -1) --out $O/bench_traffic.json > $O/pmc_traffic.log
# 4. issue / MFMA counters of the recurrent kernels alone (as round 1)
i=0
for g in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_$i -- python $R/tools/rnn_microbench.py --cell LSTM > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $(find /tmp/pmc_* -name "*counter_collection.csv") > $O/rnn_pmc_summary.txt 2>&1
# 5. the tools' own timings
python $R/tools/fit_e2e_bench.py 2>&1 | grep -v amdgpu > $O/fit_e2e.txt
python $R/tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
python $R/tools/fit_e2e_bench.py --windows 256 --songs 8 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
for args in "--config 2" "--config 5" "--config 5 --gemm-blocks 96" "--config 5 --gemm-blocks 128" "--config 5 --cell GRU"; do
  python $R/tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
python $R/tools/large_shape_check.py 2>&1 | grep -v amdgpu > $O/large_shape.txt
python $R/tools/rnn_microbench.py --cell LSTM 2>&1 | grep -v amdgpu > $O/rnn_microbench.txt
python $R/tools/rnn_microbench.py --cell GRU 2>&1 | grep -v amdgpu >> $O/rnn_microbench.txt
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_lstm_one_rank_rccl.json
ls -la $O
