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


def lab_build():
    """1 when the loaded libsbhip.so is a lab build (`make lab`, selected with
    SBHIP_LIBRARY=$PWD/sparsebench_amd/lib/lab/libsbhip.so): the product plus the measured-slower alternatives"""
    from sparsebench_amd import capi
    return bool(capi.load().sb_lab_build())


def modes_scs():
    """SpMV kernel modes to walk for an SCS C=64 matrix: the product ships 5 (masked row programs) and 0 (reference layout)"""
    return (5, 3, 2, 1, 0) if lab_build() else (5, 0)


def fused_levels():
    """sb_cg_set_fused levels to walk: the product ships 1 (five launches per body) and 0 (the reference's op list)"""
    return (1, 2, 3, 0) if lab_build() else (1, 0)


def pytest_collection_modifyitems(config, items):
    """tests marked `lab` exercise alternatives that only lab builds contain: with the product library they are DESELECTED
    (not skipped), so the default `-m gpu` suite is exactly what ships"""
    if not any(i.get_closest_marker("lab") for i in items):
        return
    try:
        lab = lab_build()
    except Exception:
        lab = False
    if lab:
        return
    keep, drop = [], []
    for i in items:
        (drop if i.get_closest_marker("lab") else keep).append(i)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: exercises measured-slower alternatives that only lab builds (`make lab`) contain; "
                                       "deselected unless SBHIP_LIBRARY points at a lab build")
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
