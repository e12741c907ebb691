cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
