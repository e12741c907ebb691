#!/bin/bash
# round 4, experiment A: where does the LSTM BPTT step go?  Ablation / filler-room variants of rnn_resident.hip (timing only).
cd "$(dirname "$0")/.."
tools/build_variants.sh base:"" nol:"-DABL_NOL=1" nox:"-DABL_NOX=1" notrg:"-DABL_NOTRG=1" nobar:"-DABL_NOBAR=1" nob:"-DABL_NOB=1" \
  nomath:"-DABL_NOMATH=1" skel:"-DABL_NOL=1 -DABL_NOX=1 -DABL_NOTRG=1 -DABL_NOB=1" \
  skelnm:"-DABL_NOL=1 -DABL_NOX=1 -DABL_NOTRG=1 -DABL_NOB=1 -DABL_NOMATH=1" \
  fill1:"-DABL_FILL=1" fill2:"-DABL_FILL=2" fill3:"-DABL_FILL=3" fill4:"-DABL_FILL=4"
