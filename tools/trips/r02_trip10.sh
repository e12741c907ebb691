cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dp_fit_gpu.py tests/test_classifier_gpu.py -m gpu -q --timeout 400 --maxfail 12 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -15 $O/gpu_tests.log | cut -c1-250
timeout 600 python tools/fit_e2e_bench.py --windows 256 --songs 8 2>&1 | grep -v amdgpu > $O/fit_e2e.txt
timeout 600 python tools/fit_e2e_bench.py --windows 256 --songs 8 --lazy 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
timeout 600 python tools/fit_e2e_bench.py --lazy 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
cat $O/fit_e2e.txt
timeout 300 python vae_training.py --epochs 3 --songs 6 --test-songs 1 2>&1 | grep -v amdgpu | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -8
