bash tools/trips/r02_trip53.sh
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02q
echo "prepass stress failures: $(wc -l < $O/stress4.txt)"
python tools/fit_e2e_bench.py 2>&1 | grep -v amdgpu > $O/fit_e2e.txt
python tools/fit_e2e_bench.py --with-prepass 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
python tools/fit_e2e_bench.py --windows 256 --songs 8 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
grep -c Traceback $O/fit_e2e.txt
timeout 1800 python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_dp_fit_gpu.py -m gpu -q --timeout 400 --maxfail 5 > $O/pytest_v.txt 2>&1
tail -3 $O/pytest_v.txt
