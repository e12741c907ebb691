cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
for i in 1 2 3; do
  rm -rf /tmp/kd$i
  MVAE_DEVICE_JOIN=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kd$i -- python tools/decode_bench.py --config 2 2>&1 | grep "decode config" | cut -c1-150 >> $O/decode_dj_trace.txt
  f=$(find /tmp/kd$i -name "*kernel_stats.csv" | head -1)
  head -8 $f | cut -c1-200 >> $O/decode_dj_trace.txt
  python tools/timeline.py $(find /tmp/kd$i -name "*kernel_trace.csv" | head -1) --min-us 30 2>/dev/null | head -60 | cut -c1-130 > $O/decode_dj_timeline_$i.txt
done
cat $O/decode_dj_trace.txt
