"""The drop-in boundary without a GPU: the C-ABI library loads, exports every symbol
include/sbhip.h declares, the per-format libraries export the reference's symbols, a
C caller written only against include/sparsebench/sparsebench.h compiles and links
for both formats, and the product never reaches into oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "sparsebench_amd", "lib")


def declared_symbols(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sb_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    from sparsebench_amd import capi
    L = capi.load()
    names = declared_symbols(os.path.join(ROOT, "include", "sbhip.h"))
    assert len(names) >= 55
    for n in names:
        assert hasattr(L, n), "libsbhip.so does not export %s" % n
    assert set(names) == set(capi.SYMBOLS), set(names) ^ set(capi.SYMBOLS)
    assert b"gfx950" in L.sb_version()
    assert L.sb_is_initialized() == 0  # loading touched no device


REF_SYMBOLS = ["convertMatrix", "spMVM", "solveCG", "waxpby", "ddot", "commInit", "commFinalize",
               "commPartition", "commDistributeMatrix", "commExchange", "commReduction",
               "commPrintBanner", "commAbort", "commBarrier", "matrixGenerate", "MMMatrixRead",
               "matrixConvertfromMM", "allocate", "getTimeStamp", "initParameter", "readParameter",
               "profilerInit", "profilerPrint", "profilerFinalize", "changeFileEnding", "getTimeResolution",
               "matrixBinRead", "matrixBinWrite", "commPrintConfig", "commGMatrixDump", "commMatrixDump",
               "commVectorDump", "printParameter"]


@pytest.mark.parametrize("fmt", ["crs", "scs"])
def test_dropin_library_exports_reference_symbols(fmt):
    from sparsebench_amd import hostapi
    hostapi.host()
    d = ctypes.CDLL(os.path.join(LIB, "libsparsebench_%s.so" % fmt))
    for s in REF_SYMBOLS:
        assert hasattr(d, s), s


@pytest.mark.parametrize("fmt", ["CRS", "SCS"])
def test_reference_shaped_caller_compiles_and_links(fmt, tmp_path):
    exe = str(tmp_path / ("dropin_driver_%s" % fmt))
    cmd = ["gcc", "-std=gnu11", "-O1", "-Wall", "-D" + fmt, "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "dropin_driver.c"), "-o", exe, "-L" + LIB,
           "-lsparsebench_%s" % fmt.lower(), "-lsparsebench_host", "-lsbhip",
           "-Wl,-rpath," + LIB, "-lm"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()
    assert os.path.exists(exe)


REF_MAIN = "/root/reference/src/main.c"
COMPAT = os.path.join(ROOT, "include", "sparsebench", "compat")


def test_compat_headers_cover_the_reference_includes():
    """one forwarding header per project include of the reference's driver (src/main.c:12-20)"""
    names = ["allocate", "comm", "matrix", "matrixBinfile", "parameter", "profiler", "solver", "timing", "util"]
    for n in names:
        assert os.path.exists(os.path.join(COMPAT, n + ".h")), n
    if os.path.exists(REF_MAIN):
        incs = re.findall(r'^#include "([A-Za-z]+)\.h"', open(REF_MAIN).read(), re.M)
        assert sorted(incs) == sorted(names)


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("fmt", ["CRS", "SCS"])
def test_reference_main_compiles_unchanged_and_links_the_dropin(fmt, tmp_path):
    """The reference's OWN driver (src/main.c), not one line changed, compiled against the forwarding
    headers and linked with the per-format drop-in library: what north_star calls "main.c still drives
    it".  (It is compiled from where it lies; nothing is copied into the repo.  oracle/build_ref.sh
    builds the same executables into oracle/_ref/ so the GPU box can RUN them: test_gpu_dropin.py.)"""
    exe = str(tmp_path / ("refmain_%s" % fmt))
    # a scratch copy: `#include "comm.h"` searches the including file's own directory first, so compiled where it
    # lies the driver would see the reference's headers (and its 16-byte serial Comm) instead of the forwarding ones
    src = str(tmp_path / "main.c")
    import shutil
    shutil.copy(REF_MAIN, src)
    assert open(src, "rb").read() == open(REF_MAIN, "rb").read()  # not one byte changed
    cmd = ["gcc", "-std=gnu11", "-O1", "-w", "-D" + fmt, "-DPRECISION=2", "-DUINT_TYPE=1", "-DARRAY_ALIGNMENT=64",
           "-I" + COMPAT, "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + LIB,
           "-lsparsebench_%s" % fmt.lower(), "-lsparsebench_host", "-lsbhip", "-Wl,-rpath," + LIB, "-lm"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()
    # every undefined symbol of the driver is satisfied by OUR libraries or libc -- none by the reference
    nm = subprocess.run(["nm", "-u", exe], stdout=subprocess.PIPE).stdout.decode()
    for sym in ("solveCG", "convertMatrix", "spMVM", "commPartition", "changeFileEnding", "matrixBinWrite"):
        assert re.search(r"\bU %s\b" % sym, nm), sym
    # the driver was built against OUR struct layouts (the reference's serial Comm is 16 bytes, ours carries the plan)
    deps = subprocess.run(["gcc", "-std=gnu11", "-w", "-D" + fmt, "-I" + COMPAT, "-I" + os.path.join(ROOT, "include"), "-M", src],
                          stdout=subprocess.PIPE).stdout.decode()
    assert "sparsebench/compat/comm.h" in deps and "/root/reference/src/comm.h" not in deps
    ldd = subprocess.run(["ldd", exe], stdout=subprocess.PIPE).stdout.decode()
    assert "libsparsebench_%s.so" % fmt.lower() in ldd and "libsbhip.so" in ldd and "sbref" not in ldd


def test_product_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "sparsebench_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="ignore").read()
                # anything that could pull oracle code in: include / import / load / link
                # lines (comments that merely mention the oracle are fine)
                for line in txt.splitlines():
                    if re.search(r"(#\s*include|\bimport\b|\bfrom\b|dlopen|CDLL|-l|-L|subprocess|exec).*"
                                 r"(oracle|sbref)", line):
                        bad.append((os.path.join(base, f), line.strip()))
    assert not bad, bad
    init = open(os.path.join(ROOT, "sparsebench_amd", "__init__.py")).read()
    assert "import oracle" not in init and "from oracle" not in init


def test_missing_library_fails_loudly(monkeypatch):
    from sparsebench_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libsbhip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load()


def test_no_device_fails_loudly():
    """On a box without a GPU the hot path must raise, not fall back."""
    from sparsebench_amd import capi
    L = capi.load()
    if L.sb_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.init(0)
