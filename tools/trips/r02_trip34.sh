cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for m in 0 1 2; do
  if [ $m != 0 ]; then touch midi-vae_amd/csrc/rnn_resident.hip; timeout 900 make -C midi-vae_amd/csrc -j8 EXTRA=-DMVAE_EXP_SAVE=$m > $O/save_build_$m.log 2>&1; fi
  echo "## saves: $m (0 plain, 1 nt, 2 write-through)" >> $O/ab_saves.txt
  timeout 300 python tools/rnn_microbench.py --cell LSTM 2>&1 | grep "fwd dense\|fwd index" >> $O/ab_saves.txt
  timeout 300 python tools/rnn_microbench.py --cell LSTM --concurrent 3 2>&1 | grep "fwd dense\|fwd index" | sed 's/^/x3 /' >> $O/ab_saves.txt
  for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line "saves=$m LSTM" >> $O/ab_saves.txt; done
  timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "saves=$m GRU" >> $O/ab_saves.txt
done
cat $O/ab_saves.txt
