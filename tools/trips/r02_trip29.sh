cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 400 --maxfail 8 > $O/pytest_full_ks.txt 2>&1
tail -8 $O/pytest_full_ks.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); print('$1', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['critical_path']['us_per_step_fwd'], d['roofline']['us_per_time_step'])"; }
for v in "1 32" "0 32" "1 32"; do
  set -- $v
  MVAE_KSTREAM_GRADS=$1 MVAE_KSTREAM_WGS=$2 timeout 600 python bench.py --no-cpu-baseline --cell GRU 2>/dev/null | line "kstream=$1 wgs=$2 GRU" >> $O/ab_ks3.txt
done
cat $O/ab_ks3.txt
