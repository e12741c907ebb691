# round 2, first GPU trip: the new parity tests + which HBM-byte counters rocprofv3 can collect on this image
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py tests/test_ops_gpu.py -k "baseline_configs or planted or resident_pipelined or config2 or config4" -x -q --durations=15 > $O/new_tests.log 2>&1
echo "pytest rc=$?" >> $O/new_tests.log
cd /tmp
timeout 120 rocprofv3 -L > $O/counters_list.txt 2>&1
grep -n -i "TCC_EA\|FETCH\|WRITE_SIZE\|HBM\|TCC_REQ\|TCC_HIT\|TCC_MISS\|MALL" $O/counters_list.txt | head -200 > $O/counters_grep.txt
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  echo "=== pass $i: $g" >> $O/pmc_try.log
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/pmct_$i -- python $R/tools/rnn_microbench.py --cell LSTM --reps 2 >> $O/pmc_try.log 2>&1
  echo "rc=$?" >> $O/pmc_try.log
  f=$(find /tmp/pmct_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_bytes.py $f >> $O/pmc_try.log 2>&1; fi
done
ls -la $O
tail -5 $O/new_tests.log
