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
    (TEMPLATE, "__device__ const double* lab_r_;\n__device__ double* lab_pnew_;\n__device__ double* lab_xs_;\n"
               "typedef const __attribute__((address_space(1))) double* gcd_t;\ntypedef __attribute__((address_space(1))) double* gd_t;\n" + TEMPLATE),
    ("  extern __shared__ __attribute__((aligned(16))) double lds[]; // [dict][exception entries + 8][window]\n",
     "  extern __shared__ __attribute__((aligned(16))) double lds[]; // [dict][exception entries + 8][window]\n"
     "  const gcd_t lab_r = (gcd_t)lab_r_;\n  const gd_t lab_pnew = (gd_t)lab_pnew_, lab_xs = (gd_t)lab_xs_;\n"),
    # variant 3 (what a product kernel would have to look like to stay within 64 VGPRs): the window in TWO halves, each
    # half loading p AND r (16 + 16 registers), p_new = r + beta p formed in registers and stored to LDS: one more
    # dependent round trip per tile instead of 30 more live registers
    ("  double t[WB];\n", "  double t[16];\n"),
    ("    for (int k = 0; k < WB; k++) t[k] = xcol(field(12 + min(k, 17)) + dmap[k]);\n",
     "    for (int k = 0; k < 8; k++) t[k] = xcol(field(12 + min(k, 17)) + dmap[k]), t[8 + k] = lab_r[field(12 + min(k, 17)) + dmap[k]];\n"),
    ("    for (int k = 0; k < WB; k++) {\n      const uint32_t slot = (uint32_t)k * 256u + threadIdx.x;\n      if (slot < win) sx[slot] = t[k];\n    }\n"
     "    if (threadIdx.x == 0) sx[0] = xpad;\n",
     "    for (int k = 0; k < 8; k++) {\n      const uint32_t slot = (uint32_t)k * 256u + threadIdx.x;\n      if (slot < win) sx[slot] = t[8 + k] + 0.5 * t[k];\n    }\n"
     "    __builtin_amdgcn_sched_barrier(0);\n    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    __builtin_amdgcn_sched_barrier(0);\n"
     "#pragma unroll\n    for (int k = 8; k < WB; k++) t[k - 8] = xcol(field(12 + min(k, 17)) + dmap[k]), t[k] = lab_r[field(12 + min(k, 17)) + dmap[k]];\n"
     "#pragma unroll\n    for (int k = 8; k < WB; k++) {\n      const uint32_t slot = (uint32_t)k * 256u + threadIdx.x;\n      if (slot < win) sx[slot] = t[k] + 0.5 * t[k - 8];\n    }\n"
     "    if (threadIdx.x == 0) sx[0] = xpad;\n"),
    # own-row operands: loaded LATE (behind the staging barrier: their latency hides behind the accumulate loop)
    ("  __syncthreads();\n  // An element costs: entry -> byte offset of its x in the window -> x -> multiply -> add.\n",
     "  __syncthreads();\n  double rown[CW], xown[CW];\n"
     "#pragma unroll\n  for (int c = 0; c < CW; c++) rown[c] = lab_r[min(row[c], nr - 1u)], xown[c] = lab_xs[min(row[c], nr - 1u)];\n"
     "  // An element costs: entry -> byte offset of its x in the window -> x -> multiply -> add.\n"),
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
     "      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sbk::lab_r_), &a, sizeof a)); HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sbk::lab_pnew_), &b, sizeof b));\n"
     "      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sbk::lab_xs_), &c, sizeof c)); } }\n"),
])
os.makedirs(ROOT + "/labs", exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
       "-Wno-unused-function", "-shared"] + sys.argv[1:] + ["-o", ROOT + "/labs/libsbhip_fusep.so", W + "/sparsebench_amd/csrc/sbhip.hip", "-ldl"]
subprocess.check_call(cmd)
print("built labs/libsbhip_fusep.so")
