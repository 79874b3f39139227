import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ns():
    import raytracetorch_amd as rta
    import scenes
    return scenes.namespace_from_package(rta)


def pytest_sessionfinish(session, exitstatus):
    """Leave the list of parity checks that needed the float64 tie-break where a GPU run can be read afterwards."""
    import json
    try:
        import util
    except ImportError:
        return
    if not util.TIE_BREAKS or os.environ.get("RTT_HIP_LIB"):
        return   # (a child session on another build of the library keeps its list to itself)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_tie_breaks.json"), "w") as fh:
            json.dump(util.TIE_BREAKS, fh, indent=1, sort_keys=True)
    except OSError:
        pass
