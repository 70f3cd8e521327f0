#!/bin/bash
# One gpurun call: GPU test suite, then (unless the suite was KILLED by its time limit) the profiling round.
# A failing assertion does not stop the call; a timeout / kill does (no further GPU step after a hung one).
TAG=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 ${2:-700} python3 -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest was killed by its time limit: stopping"; exit $rc; fi
if [ "${3:-profile}" = "profile" ]; then
  bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
  prc=$?
  echo "profile rc=$prc"; tail -8 gpurun_out/profile_round_$TAG.log
fi
exit $rc
