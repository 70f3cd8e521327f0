#!/bin/bash
# One gpurun call: GPU test suite, then (unless the suite was KILLED by its time limit) the profiling round.
# A failing assertion does not stop the call; a timeout / kill does (no further GPU step after a hung one).
TAG=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 ${2:-700} python3 -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
# go on to profiling only after pytest's own verdicts (0 = all passed, 1 = some test failed); anything else - killed by the
# time limit (124 / 137), a GPU fault or abort of the test process (134 SIGABRT, 135 SIGBUS, 139 SIGSEGV) - stops the call:
# no further GPU step after a hung or faulted one
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest ended abnormally (rc=$rc): stopping"; tail -30 gpurun_out/pytest_gpu.log; exit $rc; fi
if [ "${3:-profile}" = "profile" ]; then
  bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
  prc=$?
  echo "profile rc=$prc"; tail -8 gpurun_out/profile_round_$TAG.log
fi
exit $rc
