# what does the producer's L2 write-back (buffer_wbl2 before each chunk hand-over) cost?  timing only: the NOWB build is not valid
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "wbl2 LSTM" >> $O/ab_nowb.txt; done
timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "wbl2 GRU" >> $O/ab_nowb.txt
touch midi-vae_amd/csrc/common.h
timeout 900 make -C midi-vae_amd/csrc -j8 EXTRA=-DMVAE_EXP_NOWB > $O/nowb_build.log 2>&1
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "nowb LSTM" >> $O/ab_nowb.txt; done
timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "nowb GRU" >> $O/ab_nowb.txt
cat $O/ab_nowb.txt
