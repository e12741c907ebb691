cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
k() { python tools/knob_bench.py "$@" 2>&1 | grep -v amdgpu >> $O/knobs.txt; }
k --shape reference --cell LSTM
k --shape reference --cell LSTM defer_grads_rows=32768
k --shape reference
k --shape reference defer_grads_rows=0
k --shape reference --batch 64
k --shape reference --batch 64 defer_grads_rows=0
k --shape bench
k --shape bench --cell GRU
cut -c1-200 $O/knobs.txt
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
