cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p $R/gpurun_out/r03p
timeout 2700 python -m pytest tests -m gpu -q --timeout 600 --maxfail 10 > $R/gpurun_out/r03p/pytest_all.txt 2>&1
tail -6 $R/gpurun_out/r03p/pytest_all.txt
bash tools/collect_profiles_r03.sh > $R/gpurun_out/r03p/collect.log 2>&1
tail -5 $R/gpurun_out/r03p/collect.log
