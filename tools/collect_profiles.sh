cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01l
mkdir -p $O
python $R/bench.py > $O/bench_lstm.json 2> $O/bench_lstm.err
python $R/bench.py --cell GRU > $O/bench_gru.json 2> $O/bench_gru.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py > /dev/null 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/bench_lstm_kernel_stats.csv
python $R/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) --min-us 25 > $O/timeline_lstm_step.txt
i=0
for g in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do   # (FETCH_SIZE / WRITE_SIZE abort rocprofv3 on this image: DESIGN.md section 4)
  for c in LSTM GRU; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmc_$i -- python $R/tools/rnn_microbench.py --cell $c > /dev/null 2>&1
  done
done
python $R/tools/pmc_summary.py $(find /tmp/pmc_* -name "*counter_collection.csv") > $O/rnn_pmc_summary.txt 2>&1
ls -la $O
python $R/tools/decode_bench.py --config 2 > $O/decode.txt 2>&1
python $R/tools/decode_bench.py --config 5 >> $O/decode.txt 2>&1
python $R/tools/gemm_microbench.py > $O/gemm_microbench.txt 2>&1
python $R/tools/rnn_microbench.py --cell LSTM > $O/rnn_microbench.txt 2>&1
python $R/tools/rnn_microbench.py --cell GRU >> $O/rnn_microbench.txt 2>&1
python $R/tools/head_bench.py > $O/head_bench.txt 2>&1
MVAE_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_lstm_one_rank_rccl.json
