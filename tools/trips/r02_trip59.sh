cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
for c in 16 32 64 128; do
  MVAE_PIPE_CHUNK=$c python tools/decode_bench.py --config 5 2>&1 | grep "decode config" | cut -c1-150 | sed "s/^/chunk $c: /" >> $O/decode_chunk.txt
done
MVAE_PIPE_CHUNK=32 python tools/decode_bench.py --config 5 --cell GRU 2>&1 | grep "decode config" | cut -c1-150 | sed "s/^/chunk 32: /" >> $O/decode_chunk.txt
MVAE_PIPE_CHUNK=16 python tools/decode_bench.py --config 5 --cell GRU 2>&1 | grep "decode config" | cut -c1-150 | sed "s/^/chunk 16: /" >> $O/decode_chunk.txt
cat $O/decode_chunk.txt
