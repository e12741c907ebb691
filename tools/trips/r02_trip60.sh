cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
for c in 16 32 64; do
  MVAE_PIPE_CHUNK=$c python tools/large_shape_check.py 2>&1 | grep "ms per step" | cut -c1-90 | sed "s/^/chunk $c: /" >> $O/large_chunk.txt
done
cat $O/large_chunk.txt
