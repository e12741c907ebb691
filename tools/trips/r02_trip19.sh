# consumer acquire: system-scope invalidate (buffer_inv sc0 sc1) vs agent scope (buffer_inv sc1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "sys-inv LSTM" >> $O/ab_inv.txt; done
timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "sys-inv GRU" >> $O/ab_inv.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_full.json 2>/dev/null
touch midi-vae_amd/csrc/common.h
timeout 900 make -C midi-vae_amd/csrc -j8 EXTRA=-DMVAE_EXP_AGENT_INV > $O/inv_build.log 2>&1
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "agent-inv LSTM" >> $O/ab_inv.txt; done
timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "agent-inv GRU" >> $O/ab_inv.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 400 --maxfail 5 -k "pipelin or hand_over" > $O/pytest_agent_inv.txt 2>&1
tail -5 $O/pytest_agent_inv.txt
cat $O/ab_inv.txt
