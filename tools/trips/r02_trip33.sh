cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
for args in "" "--signal 16" "--concurrent 3" "--concurrent 3 --signal 16" "--concurrent 6"; do
  echo "## rnn_microbench.py --cell LSTM $args" >> $O/rnn_micro_conc.txt
  timeout 300 python tools/rnn_microbench.py --cell LSTM $args 2>&1 | grep -v amdgpu.ids >> $O/rnn_micro_conc.txt
done
echo "## rnn_microbench.py --cell GRU" >> $O/rnn_micro_conc.txt
timeout 300 python tools/rnn_microbench.py --cell GRU 2>&1 | grep -v amdgpu.ids >> $O/rnn_micro_conc.txt
echo "## rnn_microbench.py --cell GRU --concurrent 3 --signal 16" >> $O/rnn_micro_conc.txt
timeout 300 python tools/rnn_microbench.py --cell GRU --concurrent 3 --signal 16 2>&1 | grep -v amdgpu.ids >> $O/rnn_micro_conc.txt
cat $O/rnn_micro_conc.txt
