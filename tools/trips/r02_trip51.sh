cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
for i in 1 2 3; do
  echo "## singles=1 run $i" >> $O/prepass_repro.txt
  timeout 300 python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -v "amdgpu\|host time\|host packer" | tail -4 | cut -c1-200 >> $O/prepass_repro.txt
done
for i in 1 2; do
  echo "## singles=0 run $i" >> $O/prepass_repro.txt
  MVAE_KSTREAM_SINGLES=0 timeout 300 python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -v "amdgpu\|host time\|host packer" | tail -4 | cut -c1-200 >> $O/prepass_repro.txt
done
for i in 1 2; do
  echo "## kstream=0 run $i" >> $O/prepass_repro.txt
  MVAE_KSTREAM_GRADS=0 timeout 300 python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -v "amdgpu\|host time\|host packer" | tail -4 | cut -c1-200 >> $O/prepass_repro.txt
done
cat $O/prepass_repro.txt
