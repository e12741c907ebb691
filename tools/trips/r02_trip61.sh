cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py tests/test_engine_gpu.py tests/test_classifier_gpu.py -m gpu -q --timeout 400 --maxfail 5 > $O/pytest_chunk.txt 2>&1
tail -3 $O/pytest_chunk.txt
for args in "--config 2" "--config 5" "--config 5 --cell GRU"; do python tools/decode_bench.py $args 2>&1 | grep "decode config" | cut -c1-150; done
python tools/large_shape_check.py 2>&1 | grep "ms per step" | cut -c1-90
