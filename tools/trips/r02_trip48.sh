cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
for i in 1 2 3 4 5 6 7 8; do python tools/decode_bench.py --config 2 2>&1 | grep "decode config" | cut -c1-140 >> $O/decode_repeat.txt; done
for i in 1 2 3 4; do MVAE_DEVICE_JOIN=0 python tools/decode_bench.py --config 2 2>&1 | grep "decode config" | cut -c1-140 | sed 's/^/nojoin /' >> $O/decode_repeat.txt; done
cat $O/decode_repeat.txt
