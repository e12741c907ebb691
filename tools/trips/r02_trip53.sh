cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
for i in $(seq 1 30); do
  timeout 300 python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep "RuntimeError" | cut -c1-220 >> $O/stress4.txt
done
cat $O/stress4.txt
