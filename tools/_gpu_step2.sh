#!/bin/bash
mkdir -p gpurun_out
L=$PWD/rwm-pt-pytorch_amd
timeout -k 10 1100 bash tools/ab_bench.sh N=$L/lib_exp/libptrwm_hip.so K3=$L/lib_expK3/libptrwm_hip.so W3=$L/lib_expW3/libptrwm_hip.so K4=$L/lib_expK4/libptrwm_hip.so
