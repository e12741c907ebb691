cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
for v in 32 16 64 32 16 64; do
  MVAE_PIPE_CHUNK=$v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pipe_chunk=$v LSTM', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])" >> $O/ab_chunk.txt
done
MVAE_PIPE_CHUNK=16 timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pipe_chunk=16 GRU', d['ms_per_step'])" >> $O/ab_chunk.txt
MVAE_PIPE_CHUNK=32 timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('pipe_chunk=32 GRU', d['ms_per_step'])" >> $O/ab_chunk.txt
cat $O/ab_chunk.txt
