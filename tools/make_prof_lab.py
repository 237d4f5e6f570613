#!/usr/bin/env python3
"""make_prof_lab.py -- build labs/libsbhip_prof.so: the product library with per-tile timestamps
(wall_clock64) in spmv_scs64_pat and an extra entry point sb_lab_prof().  The sources are
patched in a scratch copy under /tmp; the repository files are not touched.  Read the result
with:  SBHIP_LIBRARY=labs/libsbhip_prof.so python tools/pat_lab.py 128 256 5"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = "/tmp/sb_prof_lab"
shutil.rmtree(W, ignore_errors=True)
os.makedirs(W + "/sparsebench_amd")
shutil.copytree(ROOT + "/include", W + "/include")
shutil.copytree(ROOT + "/sparsebench_amd/csrc", W + "/sparsebench_amd/csrc")


def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        if s.count(old) != 1:
            sys.exit("make_prof_lab: anchor not unique/present in %s:\n%s" % (path, old))
        s = s.replace(old, new)
    open(path, "w").write(s)


TEMPLATE = ("template <int CPT, bool DOT, bool SKIPPAD, bool HALO, bool MASKED>\n"
            "__global__ __launch_bounds__(256) void spmv_scs64_pat(")
patch(W + "/sparsebench_amd/csrc/pack.hip.h", [
    (TEMPLATE,
     "__device__ long long g_prof[8192 * 8];\n"
     "#define PROF(i) do { if (((PROF_POINTS) >> (i)) & 1) if (threadIdx.x == 0 && blockIdx.x < 8192) g_prof[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)\n"
     + TEMPLATE),
    ("  // A launch covers headers [firstHdr", "  PROF(0);\n  // A launch covers headers [firstHdr"),
    ("  const int stopped    = (int)field(PAT_STOP_LANE);\n", "  const int stopped    = (int)field(PAT_STOP_LANE);\n  PROF(1);\n"),
    ("  if (tile0 >= nHdrs || stopped) return; // uniform per workgroup\n",
     "  if (tile0 >= nHdrs || stopped) return; // uniform per workgroup\n  PROF(2);\n"),
    ("mine.off8 + sxOff, mine.m };\n  __syncthreads();\n", "mine.off8 + sxOff, mine.m };\n  __syncthreads();\n  PROF(3);\n"),
    ("    if (row[c] < nr) y[row[c]] = acc;\n", "    PROF(4 + c);\n    if (row[c] < nr) y[row[c]] = acc;\n"),
])
patch(W + "/sparsebench_amd/csrc/sbhip_matrix.inc.h", [
    ("uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }\n",
     "uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }\n"
     "extern \"C\" void sb_lab_prof(long long* out)\n{\n  HIP_CHECK(hipStreamSynchronize(g.stream));\n"
     "  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sbk::g_prof), sizeof(long long) * 8192 * 8));\n}\n"),
])
os.makedirs(ROOT + "/labs", exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
       "-Wno-unused-function", "-shared", "-DPROF_POINTS=" + os.environ.get("PROF_POINTS", "31")] + sys.argv[1:] + [
       "-o", ROOT + "/labs/" + os.environ.get("PROF_OUT", "libsbhip_prof.so"), W + "/sparsebench_amd/csrc/sbhip.hip", "-ldl"]
subprocess.check_call(cmd)
print("built labs/libsbhip_prof.so")
