# Round 5: full GPU suite on the current tree, the 200-step ELBO runs (4 at once: the float64 oracle is the slow side), the decisive decode report
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
export PYTHONPATH=$R
for c in LSTM GRU; do for lr in 2e-4 1e-3; do
  OMP_NUM_THREADS=24 python tests/studies/elbo_long.py --cell $c --lr $lr --steps 200 > $O/elbo_200_${c}_${lr}.txt 2> $O/elbo_200_${c}_${lr}.err &
done; done
timeout 1700 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
grep "decisive decode\|config4 decode" $O/pytest_gpu.txt > $O/decode_agreement.txt
wait
python tools/plan_host_bench.py --shape bench 2>&1 | grep -v amdgpu > $O/plan_host.txt
ls -la $O
