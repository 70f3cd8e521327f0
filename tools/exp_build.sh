#!/bin/bash
# Quick experimental build (development aid): only the RoughCarpet kernels, every other target stubbed out, into
# rwm-pt-pytorch_amd/lib_exp/ (use with PTRWM_LIB=.../lib_exp/libptrwm_hip.so python bench.py ...).
#   tools/exp_build.sh [extra hipcc flags, e.g. -DPTRWM_WAVES_SMALL=3]
set -e
cd "$(dirname "$0")/../rwm-pt-pytorch_amd/csrc"
OUT=${PTRWM_EXP_OUT:-../lib_exp}; OBJ=${PTRWM_EXP_OBJ:-../build_exp}
mkdir -p $OUT $OBJ
cat > $OBJ/stubs.hip <<'EOS'
#include "../csrc/variants.h"
namespace ptrwm {
#define STUB(fn) const TargetVariants &fn() { static const TargetVariants v = {}; return v; }
STUB(three_mixture_variants) STUB(full_rosenbrock_variants) STUB(even_rosenbrock_variants)
STUB(hybrid_rosenbrock_variants) STUB(iid_gamma_variants) STUB(iid_beta_variants) STUB(diag_gaussian_variants)
STUB(hypercube_variants) STUB(neal_funnel_variants)
}
EOS
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $*"
for f in capi.hip variants_rough_carpet.hip variants_rough_carpet2.hip $OBJ/stubs.hip; do
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $OBJ/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libptrwm_hip.so $OBJ/capi.o $OBJ/variants_rough_carpet.o $OBJ/variants_rough_carpet2.o $OBJ/stubs.o
ls -la $OUT/libptrwm_hip.so
