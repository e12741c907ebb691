cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
k() { python tools/knob_bench.py "$@" 2>&1 | grep -v amdgpu >> $O/knobs.txt; }
k --shape reference
k --shape reference defer_grads_rows=0
k --shape reference --steps 1000
k --shape reference defer_grads_rows=0 --steps 1000
python tools/plan_host_bench.py --shape reference 2>&1 | grep -v amdgpu >> $O/knobs.txt
rm -rf /tmp/ks_k; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_k -- python tools/knob_bench.py --shape reference --steps 200 >> $O/knobs.txt 2>&1
python - <<'PY' >> $O/knobs.txt
import csv,glob
f=glob.glob('/tmp/ks_k/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
ad=[r for r in rows if 'adam' in r[2]]
d=[(b[1]-a[1])/1e3 for a,b in zip(ad,ad[1:])]
d.sort()
print("under rocprofv3: %d optimizer launches; adam-to-adam us: min %.0f median %.0f p90 %.0f max %.0f" % (len(ad), d[0], d[len(d)//2], d[int(len(d)*0.9)], d[-1]))
PY
grep -v "^$" $O/knobs.txt | cut -c1-330
