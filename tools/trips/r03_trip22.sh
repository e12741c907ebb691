cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03t
mkdir -p $O
cd $R
for v in product NOSTORE NOLOAD NOMFMA; do
  if [ $v = product ]; then L=$R/midi-vae_amd/libmidivae_hip.so; else L=$R/build/variants/lib_ws_$v.so; fi
  echo "== $v" | tee -a $O/ws_ablation.txt
  MVAE_LIB=$L python tools/gemm_microbench.py 2>&1 | grep "proj_ws_k" | tee -a $O/ws_ablation.txt
done
