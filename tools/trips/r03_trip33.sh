cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03zz
mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x -k "config4 or decode or chip_filling or predict" > $O/pytest_dec.txt; timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --timeout 600 -x -k "slice_by_slice or inference_engine" >> $O/pytest_dec.txt 2>&1
tail -3 $O/pytest_dec.txt
for n in 2 4 6; do
  for args in "--config 5" "--config 5 --cell GRU"; do
    MVAE_HEAD_SLICES=$n timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 | cut -c1-170 | sed "s/^/slices=$n /" | tee -a $O/decode_ab.txt
  done
done
MVAE_HEAD_SLICES=1 python tools/decode_product_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | sed "s/^/slices=1 /" | tee -a $O/decode_ab.txt
MVAE_HEAD_SLICES=4 python tools/decode_product_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | sed "s/^/slices=4 /" | tee -a $O/decode_ab.txt
