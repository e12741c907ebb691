cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03x
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "one_hot_bottom" > $O/pytest_new.txt 2>&1
tail -15 $O/pytest_new.txt
