#!/bin/bash
# Development: timing variants of the two-waves-per-SIMD recurrent kernels (rnn_w8.hip) next to the product library, for
# tools/rnn_microbench.py via MVAE_LIB.   tools/build_w8_variants.sh name1:"-DW8_ABL_NOBAR=1 ..." name2:...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/midi-vae_amd/csrc
OUT=$ROOT/build/variants
mkdir -p $OUT
make -s -C $SRC
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMVAE_VARIANT_BUILD $flags -Rpass-analysis=kernel-resource-usage -c $SRC/rnn_w8.hip -o $OUT/w8_$name.o 2> $OUT/w8_$name.remarks &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libw8_$name.so $SRC/rnn.o $SRC/rnn_resident.o $OUT/w8_$name.o $SRC/gemm.o $SRC/heads.o $SRC/misc.o $SRC/latent.o $SRC/hostpack.o $SRC/plan.o -pthread &&
    echo built $name "scratch:" $(grep -o "ScratchSize \[bytes/lane\]: [0-9]*" $OUT/w8_$name.remarks | awk '{s+=$NF} END {print s}') ) &
done
wait
