cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
# bounded check that counter collection over bench.py terminates (value joins off when kernels are serialised), then steps 3-5
timeout -k 5 170 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_check -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 --prewarm-max 0 > $O/pmc_check.log 2>&1
rc=$?
echo "pmc check rc=$rc"
if [ $rc -eq 0 ]; then STEPS="3 4 5" timeout 1500 bash tools/collect_profiles_r03.sh > $R/gpurun_out/collect_r03_final345.log 2>&1; echo "collect rc=$?"; fi
tail -3 $O/pmc_check.log
