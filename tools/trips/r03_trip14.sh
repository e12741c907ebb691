cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 600 -x -k "gemm or proj or chunk or kstream" > $O/pytest_gemm.txt 2>&1
tail -4 $O/pytest_gemm.txt
python tools/gemm_microbench.py 2>&1 | grep -v amdgpu | tee $O/gemm_microbench.txt
for args in "--config 2" "--config 5" "--config 5 --cell GRU"; do
  timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 | tee -a $O/decode.txt
done
run() { d=$1; shift; timeout -k 5 170 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $d -- "$@" > $d.log 2>&1; echo "$d rc=$?"; }
i=0
for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  PMC="$grp" run /tmp/pmc_gemm_$i python tools/gemm_microbench.py
done
python tools/pmc_kernels.py --match "gemm|proj_ws" $(find /tmp/pmc_gemm_* -name "*counter_collection.csv") > $O/gemm_pmc_summary.txt 2>&1
grep -A4 "proj_ws_k\|gemm_fast_k<false, false" $O/gemm_pmc_summary.txt
PMC="TCC_EA0_WRREQ_sum" run /tmp/pmcb_WR python bench.py --no-cpu-baseline --steps 10 --warmup 3 --prewarm-max 0
f=$(find /tmp/pmcb_WR -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then grep "bwd_il_k" $f | cut -c1-400 > $O/pmc_TCC_EA0_WRREQ_sum_bwd_rows.csv; wc -l $O/pmc_TCC_EA0_WRREQ_sum_bwd_rows.csv; fi
for c in LSTM GRU; do timeout 300 python bench.py --no-cpu-baseline --cell $c 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$c', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench.txt; done
timeout 300 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg4', round(d['ms_per_step'],3), round(d['value']))" | tee -a $O/bench.txt
timeout 300 python bench.py --no-cpu-baseline --config 2 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg2', round(d['ms_per_step'],3), round(d['value']))" | tee -a $O/bench.txt
