#!/bin/bash
mkdir -p gpurun_out
L=$PWD/rwm-pt-pytorch_amd
args=""
for n in "" R15 R31 R15J3 R15J15 R15C; do args="$args D$n=$L/lib_exp$n/libptrwm_hip.so"; done
timeout -k 10 1100 bash tools/ab_bench.sh $args
