#!/bin/bash
# The GPU test suite as the round end runs it (one process, stops at the first failure).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=15 > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -30 gpurun_out/gpu_tests.log; exit $rc
