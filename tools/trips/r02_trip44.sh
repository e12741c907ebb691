cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in 0 192 208 160 0 224; do
  MVAE_GRAD_GEMM_BLOCKS=$v timeout 600 python bench.py --no-cpu-baseline 2>>$O/gb.err | line "grad_gemm_blocks=$v LSTM" >> $O/ab_gb.txt
done
cat $O/ab_gb.txt
