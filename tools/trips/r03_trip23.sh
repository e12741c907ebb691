cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03v
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 600 -x -k "gemm or proj or chunk or kstream" > $O/pytest_gemm.txt 2>&1
tail -4 $O/pytest_gemm.txt
for v in product NOSTORE NOLOAD NOMFMA; do
  if [ $v = product ]; then L=$R/midi-vae_amd/libmidivae_hip.so; else L=$R/build/variants/lib_ws_$v.so; fi
  echo "== $v" | tee -a $O/ws_ablation.txt
  MVAE_LIB=$L python tools/gemm_microbench.py 2>&1 | grep "proj_ws_k" | tee -a $O/ws_ablation.txt
done
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "phase_launches or pipelin" > $O/pytest_pipe.txt 2>&1
tail -3 $O/pytest_pipe.txt
for args in "--config 2" "--config 5" "--config 5 --cell GRU"; do
  timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 | tee -a $O/decode.txt
done
for c in LSTM GRU; do timeout 300 python bench.py --no-cpu-baseline --cell $c 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$c', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))" | tee -a $O/bench.txt; done
timeout 300 python bench.py --no-cpu-baseline --config 4 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg4', round(d['ms_per_step'],3), round(d['value']))" | tee -a $O/bench.txt
timeout 300 python bench.py --no-cpu-baseline --config 2 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg2', round(d['ms_per_step'],3), round(d['value']))" | tee -a $O/bench.txt
