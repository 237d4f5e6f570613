"""Content hash of the HIP sources the library is built from (sparsebench_amd/csrc/*).

bench.py's `roofline.traffic` comes from rocprofv3 --pmc passes that cannot run inside the bench process; the
committed passes (profiles/*_pmc_traffic.json, written by tools/make_pmc_traffic.py) carry this hash, and bench.py
uses an entry only when it equals the hash of the sources in the tree it runs from -- so a kernel edit invalidates
the number whether or not anybody remembered to bump a version string.
"""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_hash():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".h", ".hip")):
            continue
        h.update(name.encode())
        h.update(b"\0")
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]
