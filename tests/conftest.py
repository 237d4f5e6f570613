import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFDATA = os.path.join(GOLDEN, "ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build what is missing (hipcc cross-compiles without a GPU; seconds)
    need = [os.path.join(ROOT, "sparsebench_amd", "lib", n)
            for n in ("libsbhip.so", "libsparsebench_host.so", "libsparsebench_crs.so",
                      "libsparsebench_scs.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "hip", "host"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"],
                              stdout=subprocess.DEVNULL)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_1rank():
    return load_json("cg_hist_1rank.json")


@pytest.fixture(scope="session")
def golden_mpi():
    return load_json("cg_hist_mpi.json")


@pytest.fixture(scope="session")
def gpu():
    """Initialise the HIP layer once; fails loudly when there is no device."""
    from sparsebench_amd import capi
    L = capi.init(0)
    yield L
