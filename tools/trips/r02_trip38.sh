cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 400 --maxfail 8 > $O/pytest_full.txt 2>&1
tail -4 $O/pytest_full.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -3 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
