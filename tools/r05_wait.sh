cd /root/repo
for lib in "" build/variants/lib_occ2.so; do
  echo "== MVAE_LIB=$lib"
  MVAE_LIB=$lib timeout 300 python tools/gemm_microbench.py 2>&1 | grep -E "dU|kstream|dx ="
  MVAE_LIB=$lib timeout 300 python tools/knob_bench.py --shape bench --steps 60 2>&1 | tail -1 | cut -c1-150
  MVAE_LIB=$lib timeout 300 python tools/knob_bench.py --shape bench --cell GRU --steps 60 2>&1 | tail -1 | cut -c1-150
  MVAE_LIB=$lib timeout 300 python tools/knob_bench.py --shape reference --steps 200 2>&1 | tail -1 | cut -c1-150
done
