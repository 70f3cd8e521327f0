#!/bin/bash
mkdir -p gpurun_out
L=$PWD/rwm-pt-pytorch_amd
echo "== cfg4 (EvenRosenbrock dim 30 + Laplace)"
AB_ARGS="--workload cfg4 --inner 500" timeout -k 10 500 bash tools/ab_bench.sh ER1=$L/lib_expER1/libptrwm_hip.so ER7=$L/lib_expER7/libptrwm_hip.so ER15=$L/lib_expER15/libptrwm_hip.so
