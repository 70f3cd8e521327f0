#!/bin/bash
# Shows, on the GPU, that the production-path parity tests catch a slipped Philox word (VERDICT r02 next #1 "Done"): builds
# the RoughCarpet kernels with -DPTRWM_INJECT_ACCEPT_WORD_SLIP (csrc/proposals.h: the Normal proposal's accept uniform is
# taken one pair early in the one-thread-per-replica kernel) into rwm-pt-pytorch_amd/lib_inject/ and runs the Philox-mode
# tests of the RoughCarpet / Normal family against it.  The thread-form cases must FAIL (a wrong Metropolis decision, named
# with its step and temperature), the lane-split cases - whose kernel does not have the injected fault - must pass.
#   step 1 (anywhere):   tools/inject_slip_check.sh build
#   step 2 (GPU box):    tools/inject_slip_check.sh run > profiles/r03_injected_slip.txt
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = "build" ]; then
  PTRWM_EXP_OUT=../lib_inject PTRWM_EXP_OBJ=../build_inject bash tools/exp_build.sh -mllvm -enable-post-misched=0 \
      -mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize -DPTRWM_INJECT_ACCEPT_WORD_SLIP
  exit $?
fi
LIB=$PWD/rwm-pt-pytorch_amd/lib_inject/libptrwm_hip.so
echo "# tools/inject_slip_check.sh run: library built with -DPTRWM_INJECT_ACCEPT_WORD_SLIP (thread-form Normal proposal: accept word - 2)"
echo "# expected: the 'thread' cases FAIL with a proven-wrong Metropolis decision, the 'quad' cases pass"
PTRWM_LIB=$LIB python -m pytest tests/test_gpu_engine_parity.py -q -m gpu -k "test_philox_mode_vs_oracle and rc15_d30" 2>&1 \
  | grep -E "WRONG Metropolis|AssertionError: ladder|passed|failed|^FAILED|^PASSED" | cut -c1-260
echo "# the same tests against the shipped library:"
python -m pytest tests/test_gpu_engine_parity.py -q -m gpu -k "test_philox_mode_vs_oracle and rc15_d30" 2>&1 | grep -E "passed|failed"
