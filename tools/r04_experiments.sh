#!/bin/bash
# Round 4: the commands behind profiles/r04_[a-l]_* (one section per experiment; run a section on the GPU box with
#   gpurun -- 'bash tools/r04_experiments.sh <letter>').  Variant libraries are built HERE (no GPU needed) by tools/build_variants.sh
#   into build/variants/ and travel with the snapshot.  The full evidence set of the round is tools/collect_profiles_r04.sh.
cd "$(dirname "$0")/.."
mb() { MVAE_LIB=$PWD/build/variants/lib_$1.so python tools/rnn_microbench.py --cell LSTM --reps 8 2>&1 | grep "${2:-bwd\|fwd dense\|fwd const }"; }
case "$1" in
build)  # every variant the sections below use
  tools/build_variants.sh base:"" nol:"-DABL_NOL=1" nox:"-DABL_NOX=1" notrg:"-DABL_NOTRG=1" nobar:"-DABL_NOBAR=1" nob:"-DABL_NOB=1" \
    nomath:"-DABL_NOMATH=1" skel:"-DABL_NOL=1 -DABL_NOX=1 -DABL_NOTRG=1 -DABL_NOB=1" \
    skelnm:"-DABL_NOL=1 -DABL_NOX=1 -DABL_NOTRG=1 -DABL_NOB=1 -DABL_NOMATH=1" \
    fill1:"-DABL_FILL=1" fill2:"-DABL_FILL=2" fill3:"-DABL_FILL=3" fill4:"-DABL_FILL=4" \
    skew1:"-DRES_WSKEW=1" skew2:"-DRES_WSKEW=2" skew3:"-DRES_WSKEW=3" skew4:"-DRES_WSKEW=4" skew6:"-DRES_WSKEW=6" \
    olddhs:"-DBWL_OLD_DHS=1" v01:"-DBWL_ASM_1MSQ=0 -DBWL_MFMA_FIRST=1" v10:"-DBWL_MFMA_FIRST=0" v00:"-DBWL_ASM_1MSQ=0 -DBWL_MFMA_FIRST=0" \
    v11:"-DBWL_MFMA_FIRST=1" xa:"-DABL_NOX=1 -DABL_NOTRG=1" xb:"-DABL_NOX=1 -DABL_NOTRG=1 -DABL_NOBAR=1" \
    nb1:"-DABL_NOBAR1=1" nb2:"-DABL_NOBAR2=1" nb12:"-DABL_NOBAR1=1 -DABL_NOBAR2=1" ;;
a)  # where the LSTM BPTT step goes: whole-kernel timing ablations, filler room of the MFMA phase -> r04_a_bptt_ablation.txt
  for v in base nol nox notrg nobar nob nomath skel skelnm fill1 fill2 fill3 fill4; do echo "## $v"; mb $v; done ;;
b)  # wave skew behind the barrier -> r04_b_wave_skew.txt
  for v in base skew1 skew2 skew3 skew4 skew6; do echo "## $v"; mb $v; done ;;
c)  # one-clamp gate select against the old derivative; host enqueue with and without plans -> r04_c_*
  for v in olddhs; do echo "## $v"; mb $v bwd; done; echo "## product"; python tools/rnn_microbench.py --cell LSTM --reps 8 | grep bwd
  for s in bench reference; do python tools/plan_host_bench.py --shape $s; done; python tools/plan_host_bench.py --shape reference --cell LSTM ;;
f)  # what slows the recurrences inside the step -> r04_f_in_step_ab.txt, r04_in_step_probe.txt
  for e in MVAE_X=0 MVAE_PIPE_GEMM_BLOCKS=32 MVAE_PIPE_GEMM_BLOCKS=16 MVAE_KSTREAM_GRADS=0 "MVAE_DIAG_NO_PARAM_GRADS=1 MVAE_KSTREAM_GRADS=0"; do
    echo "## $e"; env $e python bench.py --no-cpu-baseline --no-other-configs --steps 24 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.3f  bwd us/step %.3f  fwd us/step %.3f' % (d['ms_per_step'], r['us_per_time_step'], r['critical_path']['us_per_step_fwd']))"; done
  python tools/in_step_probe.py ;;
h)  # operand-modifier 1 - y^2 / zero-C first MFMA, separately -> r04_h_bptt_ab.txt   (vXY: X = BWL_ASM_1MSQ, Y = BWL_MFMA_FIRST)
  for v in v00 v01 v10 v11 v00 v11; do echo "## $v"; mb $v bwd; done ;;
j)  # barriers with and without the memory instructions; copies of the launch side by side -> r04_j_*
  for v in xa xb nobar; do echo "## $v"; mb $v; done
  for args in "" "--signal 16" "--concurrent 3" "--concurrent 3 --signal 16" "--concurrent 2" "--concurrent 6"; do
    echo "## product $args"; python tools/rnn_microbench.py --cell LSTM --reps 8 $args | grep "bwd\|fwd dense\|fwd const "; done ;;
k)  # each barrier alone -> r04_k_bptt_single_barriers.txt; paired gather -> r04_k_lstm_fwd_index_paired.txt
  for v in nb1 nb2 nb12; do echo "## $v"; mb $v bwd; done; python tools/rnn_microbench.py --cell LSTM --reps 8 | grep fwd ;;
l)  # GRU paired gather; written-out table rows against the paired gather -> r04_l_*
  python tools/rnn_microbench.py --cell GRU --reps 8 | grep fwd
  for c in GRU LSTM; do for e in MVAE_INDEX_DENSE=0 MVAE_INDEX_DENSE=1; do echo "## $c $e"
    env $e python bench.py --cell $c --no-cpu-baseline --no-other-configs --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step %.3f value %.0f' % (d['ms_per_step'], d['value']))"; done; done ;;
*) echo "usage: $0 build|a|b|c|f|h|j|k|l" ;;
esac
