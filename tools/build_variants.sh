#!/bin/bash
# Development: build timing variants of the resident recurrent kernels (ablations / register configs) next to the
# product library, for tools/rnn_microbench.py via MVAE_LIB.   tools/build_variants.sh name1:"-DFLAG=1 ..." name2:...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/midi-vae_amd/csrc
OUT=$ROOT/build/variants
mkdir -p $OUT
make -s -C $SRC
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMVAE_VARIANT_BUILD $flags -c $SRC/rnn_resident.hip -o $OUT/rr_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $SRC/rnn.o $OUT/rr_$name.o $SRC/rnn_w8.o $SRC/gemm.o $SRC/heads.o $SRC/misc.o $SRC/latent.o $SRC/hostpack.o $SRC/plan.o -pthread &&
    echo built $name ) &
done
wait
