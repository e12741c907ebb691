cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
timeout 300 python tools/host_vs_device.py 2>&1 | grep -v amdgpu | tee $O/host_vs_device.txt
timeout 300 python tools/host_vs_device.py --cell GRU 2>&1 | grep -v amdgpu | tee -a $O/host_vs_device.txt
