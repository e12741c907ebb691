cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
k() { python tools/knob_bench.py "$@" 2>&1 | grep -v amdgpu >> $O/knobs.txt; }
for c in GRU LSTM; do
k --shape reference --cell $c
k --shape reference --cell $c defer_grads_rows=0
done
k --shape reference --cell GRU --batch 64
k --shape reference --cell GRU --batch 64 defer_grads_rows=0
k --shape bench
k --shape bench --cell GRU
cat $O/knobs.txt; tail -5 $O/pytest_gpu.txt
