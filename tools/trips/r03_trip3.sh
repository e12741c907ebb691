cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/anyorder_probe.hip -o /tmp/anyorder_probe && timeout 60 /tmp/anyorder_probe > $O/anyorder_probe.txt 2>&1
cat $O/anyorder_probe.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['value']))"; }
for rep in 1 2; do
for m in 0 1 2; do
  MVAE_PREP_INPUTS=$m timeout 600 python bench.py --no-cpu-baseline 2>>$O/bench.err | line "prep_inputs=$m LSTM" | tee -a $O/ab_prep_inputs.txt
done
done
for m in 0 1; do
  MVAE_PREP_INPUTS=$m timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>>$O/bench.err | line "prep_inputs=$m GRU" | tee -a $O/ab_prep_inputs.txt
done
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5 2>>$O/bench.err | line "cfg4" | tee -a $O/cfg4.txt; done
timeout 300 python tools/decode_bench.py --config 5 2>&1 | grep -v amdgpu | tee -a $O/cfg4.txt
