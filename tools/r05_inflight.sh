cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
k() { python tools/knob_bench.py "$@" 2>&1 | grep -v amdgpu >> $O/knobs.txt; }
for sh in "reference" "reference --cell LSTM" "bench" "bench --cell GRU"; do
k --shape $sh
k --shape $sh gate_pipe_gemms=0
k --shape $sh --inflight 1
k --shape $sh gate_pipe_gemms=0 --inflight 1
done
cut -c1-250 $O/knobs.txt
