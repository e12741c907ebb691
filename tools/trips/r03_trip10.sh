cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['value']), round(d['roofline']['us_per_time_step'],3))"; }
for rep in 1 2; do
for c in LSTM GRU; do
for m in 0 1; do
  MVAE_HOLD_DEC_GRADS=$m timeout 600 python bench.py --no-cpu-baseline --cell $c 2>>$O/bench.err | line "hold_dec_grads=$m $c" | tee -a $O/ab_hold.txt
done
done
done
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5 2>>$O/bench.err | line "cfg4" | tee -a $O/cfg.txt; done
timeout 600 python bench.py --no-cpu-baseline --config 2 --steps 10 --warmup 3 2>>$O/bench.err | line "cfg2" | tee -a $O/cfg.txt
MVAE_PHASE_MULTI=0 timeout 600 python bench.py --no-cpu-baseline --config 2 --steps 10 --warmup 3 2>>$O/bench.err | line "cfg2 multi=0" | tee -a $O/cfg.txt
MVAE_PHASE_MULTI=0 timeout 600 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5 2>>$O/bench.err | line "cfg4 multi=0" | tee -a $O/cfg.txt
for n in 256 1024; do
  timeout 600 python tools/fit_e2e_bench.py --songs 4 --windows $n --with-prepass 2>&1 | grep -v amdgpu | tail -4 | tee -a $O/fit_e2e.txt
done
tail -3 $O/bench.err
