#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in retrace adjoint; do
  RTT_HIP_LIB=$PWD/raytracetorch_amd/csrc/librtt_hip_prof.so RTT_BACKWARD=$mode timeout -k 10 120 python tools/phase_timers.py || exit 1
done
