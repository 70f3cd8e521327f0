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
#define STUB(fn) const TargetVariants &fn##_narrow() { static const TargetVariants v = {}; return v; } \
                 const TargetVariants &fn##_wide() { static const TargetVariants v = {}; return v; } \
                 const QuadVariants &fn##_quad() { static const QuadVariants v = {}; return v; }
STUB(three_mixture_variants) STUB(three_mixture1_variants) STUB(full_rosenbrock_variants) STUB(even_rosenbrock_variants)
STUB(hybrid_rosenbrock_variants) STUB(iid_gamma_variants) STUB(iid_beta_variants) STUB(diag_gaussian_variants)
STUB(hypercube_variants) STUB(neal_funnel_variants)
}
EOS
# the flags given on the command line go to the NARROW width group (the one the Makefile's SCHED applies to); the WIDE
# group and capi.hip are built with the defaults, as in the Makefile
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $PTRWM_EXP_ALL_FLAGS"  # PTRWM_EXP_ALL_FLAGS: every unit
/opt/rocm/bin/hipcc $BASE -c capi.hip -o $OBJ/capi.o &
/opt/rocm/bin/hipcc $BASE -c $OBJ/stubs.hip -o $OBJ/stubs.o &
for v in rough_carpet rough_carpet2; do
  /opt/rocm/bin/hipcc $BASE $* -c variants_$v.hip -o $OBJ/variants_$v.o &
  /opt/rocm/bin/hipcc $BASE $PTRWM_EXP_WIDE_FLAGS -DPTRWM_PART_WIDE -c variants_$v.hip -o $OBJ/variants_$v.wide.o &
  /opt/rocm/bin/hipcc $BASE -fno-slp-vectorize $PTRWM_EXP_QUAD_FLAGS -c quad_$v.hip -o $OBJ/quad_$v.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libptrwm_hip.so $OBJ/capi.o $OBJ/stubs.o \
  $OBJ/variants_rough_carpet.o $OBJ/variants_rough_carpet.wide.o $OBJ/variants_rough_carpet2.o $OBJ/variants_rough_carpet2.wide.o \
  $OBJ/quad_rough_carpet.o $OBJ/quad_rough_carpet2.o
ls -la $OUT/libptrwm_hip.so
