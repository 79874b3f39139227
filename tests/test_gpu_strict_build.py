"""GPU: the benchmark configurations' goldens through the strict build of the kernel library (the same source
compiled with -ffp-contract=off and correctly rounded divide / sqrt, i.e. the reference's operation order), loaded
through RTT_HIP_LIB in a child process.  Separates rounding from logic: both builds must reproduce the reference
within the plain tolerance."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRICT = os.path.join(ROOT, "raytracetorch_amd", "csrc", "librtt_hip_strict.so")

CASES = ["tests/test_gpu_parity.py::test_primitive_forward[plane_c1]",
         "tests/test_gpu_parity.py::test_sequential_trace[singlet]",
         "tests/test_gpu_parity.py::test_sequential_trace[system5]",
         "tests/test_gpu_parity.py::test_naive_ray_cast_and_step",
         "tests/test_gpu_parity.py::test_spot_size_loss_value"]


def test_benchmark_goldens_through_the_strict_build():
    assert os.path.exists(STRICT), "librtt_hip_strict.so is not built (make -C raytracetorch_amd/csrc)"
    env = dict(os.environ, RTT_HIP_LIB=STRICT)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + CASES, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"{len(CASES)} passed" in r.stdout
