cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3))"; }
for v in 0 128 192 64 0 128; do
  MVAE_DEC_GEMM_BLOCKS=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line "LSTM dec_gemm_blocks=$v" >> $O/ab_decmb.txt
done
cat $O/ab_decmb.txt
