cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03y
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 600 -x -k "rnn or bptt or bwd or lstm" > $O/pytest_rnn.txt 2>&1
tail -3 $O/pytest_rnn.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches or elbo or oracle" > $O/pytest_eng.txt 2>&1
tail -3 $O/pytest_eng.txt
for v in product novm; do
  if [ $v = product ]; then L=$R/midi-vae_amd/libmidivae_hip.so; else L=$R/build/variants/lib_$v.so; fi
  echo "== $v" | tee -a $O/ab.txt
  MVAE_LIB=$L python tools/rnn_microbench.py --cell LSTM 2>&1 | grep "bwd" | tee -a $O/ab.txt
  for rep in 1 2; do
    MVAE_LIB=$L timeout 300 python bench.py --no-cpu-baseline --cell LSTM 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('LSTM $v', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/ab.txt
  done
done
