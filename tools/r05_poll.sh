cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R; V=$R/build/variants
k() { python tools/knob_bench.py "$@" 2>&1 | grep -v amdgpu | sed "s/^/[$TAG] /" >> $O/knobs.txt; }
for rep in 1 2; do
for v in 32 8 127; do
  export TAG="poll sleep $v"; if [ $v = 32 ]; then unset MVAE_LIB; else export MVAE_LIB=$V/lib_poll$v.so; fi
  k --shape bench
  k --shape bench --cell GRU
  k --shape reference
done; done
unset MVAE_LIB
for a in "--config 4 --steps 20 --warmup 5" "--config 2 --steps 10 --warmup 3"; do
for v in 32 8; do if [ $v = 32 ]; then unset MVAE_LIB; else export MVAE_LIB=$V/lib_poll$v.so; fi
echo "[poll sleep $v] bench.py $a: $(python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")" >> $O/knobs.txt
done; done
cut -c1-170 $O/knobs.txt
