#!/usr/bin/env python3
"""make_fusep_lab.py -- FEASIBILITY lab (round 3): what would the masked pattern SpMV cost if it also took the p update
(p = r + beta p, x += alpha p_old; src/CGSolver.c:114,127) -- i.e. if the 12.5 us cg_update_p launch disappeared into the
SpMV's window staging?  Builds labs/libsbhip_fusep.so from a scratch copy of the sources (repository files untouched):
the mapped-window staging gets a second pass (the r window through the same slot map, LDS read-modify-write
sx = r + beta sx), and every row additionally loads r, x of its own row and stores x and p_new.  TIMING ONLY: the extra
operands are scratch buffers, results are not meaningful.  Read with
    SBHIP_LIBRARY=labs/libsbhip_fusep.so python tools/pat_lab.py 128 256 5"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = "/tmp/sb_fusep_lab"
shutil.rmtree(W, ignore_errors=True)
os.makedirs(W + "/sparsebench_amd")
shutil.copytree(ROOT + "/include", W + "/include")
shutil.copytree(ROOT + "/sparsebench_amd/csrc", W + "/sparsebench_amd/csrc")


def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        if s.count(old) != 1:
            sys.exit("make_fusep_lab: anchor not unique/present in %s:\n%s" % (path, old))
        s = s.replace(old, new)
    open(path, "w").write(s)


TEMPLATE = ("template <int CPT, bool DOT, bool SKIPPAD, bool HALO, bool MASKED>\n"
            "__global__ __launch_bounds__(256) void spmv_scs64_pat(")
patch(W + "/sparsebench_amd/csrc/pack.hip.h", [
    (TEMPLATE, "__device__ const double* lab_r;\n__device__ double* lab_pnew;\n__device__ double* lab_xs;\n" + TEMPLATE),
    # own-row operands, early (next to the row bases)
    ("    xrow[c]  = DOT ? x[min(row[c], nr - 1u)] : 0.0;\n  }\n",
     "    xrow[c]  = DOT ? x[min(row[c], nr - 1u)] : 0.0;\n  }\n"
     "  double rown[CW], xown[CW];\n"
     "#pragma unroll\n  for (int c = 0; c < CW; c++) rown[c] = lab_r[min(row[c], nr - 1u)], xown[c] = lab_xs[min(row[c], nr - 1u)];\n"),
    ("  for (int c = 0; c < CW; c++) asm volatile(\"\" ::\"v\"(base[c]), \"v\"(xrow[c]));\n",
     "  for (int c = 0; c < CW; c++) asm volatile(\"\" ::\"v\"(base[c]), \"v\"(xrow[c]), \"v\"(rown[c]), \"v\"(xown[c]));\n"),
    # second staging pass
    ("    if (threadIdx.x == 0) sx[0] = xpad;\n    if (win > 256u * WB) __builtin_trap(); // (the host builds no such window)\n",
     "    if (threadIdx.x == 0) sx[0] = xpad;\n    if (win > 256u * WB) __builtin_trap(); // (the host builds no such window)\n"
     "    { // LAB: the r window through the same map, then sx = r + beta sx (each thread its own slots: no barrier)\n"
     "      const double beta = 0.5;\n"
     "#pragma unroll\n      for (int k = 0; k < WB; k++) t[k] = lab_r[field(12 + min(k, 17)) + dmap[k]];\n"
     "#pragma unroll\n      for (int k = 0; k < WB; k++) {\n        const uint32_t slot = (uint32_t)k * 256u + threadIdx.x;\n"
     "        if (slot < win) sx[slot] = t[k] + beta * sx[slot];\n      }\n    }\n"),
    ("    if (row[c] < nr) y[row[c]] = acc;\n",
     "    if (row[c] < nr) y[row[c]] = acc;\n"
     "    if (row[c] < nr) { lab_xs[row[c]] = xown[c] + 0.25 * xrow[c]; const double pn = rown[c] + 0.5 * xrow[c]; lab_pnew[row[c]] = pn; xrow[c] = pn; }\n"),
])
patch(W + "/sparsebench_amd/csrc/sbhip_launch.inc.h", [
    ("  if (masked && !pm->mHdrs) SB_FATAL(\"the matrix has no masked row programs\");\n",
     "  if (masked && !pm->mHdrs) SB_FATAL(\"the matrix has no masked row programs\");\n"
     "  { static bool once = false; if (!once) { once = true; double *a, *b, *c; size_t nb = ((size_t)pm->nc + 1024) * 8;\n"
     "      HIP_CHECK(hipMalloc(&a, nb)); HIP_CHECK(hipMalloc(&b, nb)); HIP_CHECK(hipMalloc(&c, nb));\n"
     "      HIP_CHECK(hipMemset(a, 0, nb)); HIP_CHECK(hipMemset(b, 0, nb)); HIP_CHECK(hipMemset(c, 0, nb));\n"
     "      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sbk::lab_r), &a, sizeof a)); HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sbk::lab_pnew), &b, sizeof b));\n"
     "      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sbk::lab_xs), &c, sizeof c)); } }\n"),
])
os.makedirs(ROOT + "/labs", exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
       "-Wno-unused-function", "-shared"] + sys.argv[1:] + ["-o", ROOT + "/labs/libsbhip_fusep.so", W + "/sparsebench_amd/csrc/sbhip.hip", "-ldl"]
subprocess.check_call(cmd)
print("built labs/libsbhip_fusep.so")
