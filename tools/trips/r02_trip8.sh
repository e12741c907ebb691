cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 400 --maxfail 12 --durations=5 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -25 $O/gpu_tests.log | cut -c1-250
timeout 600 python tools/fit_e2e_bench.py 2>&1 | grep -v amdgpu > $O/fit_e2e.txt
timeout 600 python tools/fit_e2e_bench.py --windows 256 --songs 8 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
timeout 600 python tools/fit_e2e_bench.py --windows 256 --songs 8 --with-prepass 2>&1 | grep -v amdgpu >> $O/fit_e2e.txt
cat $O/fit_e2e.txt
for args in "--config 5" "--config 5 --gemm-blocks 128" "--config 5 --gemm-blocks 112"; do
  timeout 300 python tools/decode_bench.py $args 2>&1 | grep -v amdgpu | head -1 >> $O/decode.txt
done
cat $O/decode.txt
timeout 300 python vae_training.py --epochs 2 --songs 4 --test-songs 1 2>&1 | grep -v amdgpu | tail -5
timeout 300 python style_classifier_training.py --kind pitch --epochs 2 --songs 6 --test-songs 2 2>&1 | grep -v amdgpu | tail -3
timeout 300 python style_classifier_training.py --kind velocity --epochs 2 --songs 6 --test-songs 2 2>&1 | grep -v amdgpu | tail -2
timeout 300 python style_classifier_training.py --kind instrument --epochs 2 --songs 6 --test-songs 2 2>&1 | grep -v amdgpu | tail -2
