cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
run() { d=$1; shift; timeout -k 5 170 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $d -- "$@" > $d.log 2>&1; echo "$d rc=$?"; }
PMC="WRITE_SIZE" run /tmp/pmcb_WRITE_SIZE python bench.py --no-cpu-baseline --steps 10 --warmup 3 --prewarm-max 0
f=$(find /tmp/pmcb_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then grep "bwd_il_k" $f | cut -c1-400 > $O/pmc_WRITE_SIZE_bwd_rows.csv; fi
PMC="FETCH_SIZE" run /tmp/pmcb_FETCH_SIZE python bench.py --no-cpu-baseline --steps 10 --warmup 3 --prewarm-max 0
g=$(find /tmp/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$g" ]; then grep "bwd_il_k" $g | cut -c1-400 > $O/pmc_FETCH_SIZE_bwd_rows.csv; fi
if [ -n "$f" ] && [ -n "$g" ]; then python tools/pmc_traffic.py --fetch $g --write $f --out $O/bench_traffic.json > $O/pmc_traffic.log 2>&1; fi
i=0
for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  PMC="$grp" run /tmp/pmc_gemm_$i python tools/gemm_microbench.py
  PMC="$grp" run /tmp/pmc_dec_$i python tools/decode_bench.py --config 5 --reps 1
  PMC="$grp" run /tmp/pmc_rnn_$i python tools/rnn_microbench.py --cell LSTM
done
python tools/pmc_kernels.py --match "gemm|proj_ws" $(find /tmp/pmc_gemm_* -name "*counter_collection.csv") > $O/gemm_pmc_summary.txt 2>&1
python tools/pmc_kernels.py --match "proj_ws|fwd_il_k|fwd_multi|head_k" $(find /tmp/pmc_dec_* -name "*counter_collection.csv") > $O/decode_pmc_summary.txt 2>&1
python tools/pmc_summary.py $(find /tmp/pmc_rnn_* -name "*counter_collection.csv") > $O/rnn_pmc_summary.txt 2>&1
timeout 300 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -q -s -k "agreement" 2>&1 | grep -v amdgpu | tail -8 > $O/decode_bf16_agreement.txt
cat $O/decode_bf16_agreement.txt
# the tools' own timings with the final code
python tools/gemm_microbench.py 2>&1 | grep -v amdgpu > $O/gemm_microbench.txt
python tools/rnn_microbench.py --cell LSTM 2>&1 | grep -v amdgpu > $O/rnn_microbench.txt
python tools/rnn_microbench.py --cell GRU 2>&1 | grep -v amdgpu >> $O/rnn_microbench.txt
rm -f $O/fit_e2e.txt
for args in "" "--with-prepass" "--windows 256 --songs 8" "--windows 256 --songs 8 --with-prepass" "--with-prepass --lazy"; do
  echo "== tools/fit_e2e_bench.py $args" >> $O/fit_e2e.txt
  timeout 300 python tools/fit_e2e_bench.py $args 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
done
rm -f $O/decode.txt
for args in "--config 2" "--config 5" "--config 5 --cell GRU"; do
  timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
timeout 300 python tools/decode_product_bench.py 2>&1 | grep -v amdgpu >> $O/decode.txt
timeout 600 python tools/large_shape_check.py 2>&1 | grep -v amdgpu > $O/large_shape.txt
for m in 0 1 2; do
  MVAE_HOLD_DEC_GRADS=$m timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('hold_dec_grads=$m LSTM', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/ab_hold2.txt
done
ls $O
