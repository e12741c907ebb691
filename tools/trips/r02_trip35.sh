cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
bash tools/build_variants.sh stamps:"-DRES_STAMPS=1" > $O/stamps_build.log 2>&1
export MVAE_LIB=$R/build/variants/lib_stamps.so
for c in 1 3 6; do
  for w in fwd bwd; do
    echo "## LSTM $w dense, $c concurrent" >> $O/stamps.txt
    timeout 120 python tools/rnn_stamps.py --cell LSTM --which $w --concurrent $c 2>&1 | grep -v amdgpu.ids >> $O/stamps.txt
  done
done
cat $O/stamps.txt
