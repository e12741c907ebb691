cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x -k "attach or twenty or nonzero or phase_launches" > $O/pytest_attach.txt 2>&1
tail -25 $O/pytest_attach.txt
