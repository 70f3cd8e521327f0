#!/bin/bash
# One more experimental library that differs from rwm-pt-pytorch_amd/lib_exp (tools/exp_build.sh) in the headline translation
# unit only (development aid):   tools/exp_variant.sh <name> [hipcc flags for variants_rough_carpet2.hip]
# -> rwm-pt-pytorch_amd/lib_exp<name>/libptrwm_hip.so
set -e
cd "$(dirname "$0")/../rwm-pt-pytorch_amd/csrc"
n=$1; shift
OBJ=../build_exp; OUT=../lib_exp$n
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -enable-post-misched=0 \
  -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize "$@" -c variants_rough_carpet2.hip -o $OBJ/variants_rough_carpet2.$n.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libptrwm_hip.so $OBJ/capi.o $OBJ/stubs.o \
  $OBJ/variants_rough_carpet.o $OBJ/variants_rough_carpet.wide.o $OBJ/variants_rough_carpet2.$n.o $OBJ/variants_rough_carpet2.wide.o \
  $OBJ/quad_rough_carpet.o $OBJ/quad_rough_carpet2.o
python3 ../../tools/kernel_stats.py --objs $OBJ/variants_rough_carpet2.$n.o 2>/dev/null | grep "NormalProposal<30>, 30, true, false, false>" | sed "s/^/$n: /" | cut -c1-40,100-200
