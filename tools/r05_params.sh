cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_model_gpu.py tests/test_dp_fit_gpu.py -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
python tools/training_script_bench.py --epochs 5 2>&1 | grep -v amdgpu > $O/training_script_default.txt; cat $O/training_script_default.txt
