#!/usr/bin/env python3
"""make_scalar_lab.py -- build labs/libsbhip_scalar.so: the product library with wall_clock64 stamps in the CG scalar
step (cg_scalar_k: kernel entry, partial loads back, reduction done, stores issued) and an entry point sb_lab_sprof().
VERDICT r2 item 8: split the ~4.5 us of a scalar step into dispatch, memory round trip, reduction, stores and teardown --
the stamps give the in-kernel part, rocprofv3's begin / end timestamps of the same launches (tools/scalar_anatomy.py) the
rest.  Sources patched in a scratch copy under /tmp; the repository files are not touched."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = "/tmp/sb_scalar_lab"
shutil.rmtree(W, ignore_errors=True)
os.makedirs(W + "/sparsebench_amd")
shutil.copytree(ROOT + "/include", W + "/include")
shutil.copytree(ROOT + "/sparsebench_amd/csrc", W + "/sparsebench_amd/csrc")


def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        if s.count(old) != 1:
            sys.exit("make_scalar_lab: anchor not unique/present in %s:\n%s" % (path, old))
        s = s.replace(old, new)
    open(path, "w").write(s)


patch(W + "/sparsebench_amd/csrc/kernels.hip.h", [
    ("__device__ __forceinline__ double level1(const double* __restrict__ q, uint32_t i)\n",
     "__device__ long long g_sprof[4096 * 4];\n__device__ unsigned int g_sprof_n;\n__device__ long long g_sprof_t1;\n"
     "__device__ __forceinline__ double level1(const double* __restrict__ q, uint32_t i)\n"),
    # loads back: the per-thread sum s is complete (all of this thread's loads have returned)
    ("  s = butterfly64(s);\n  if ((threadIdx.x & 63u) == 0) lds16[threadIdx.x >> 6] = s;\n  __syncthreads();\n  double total = lds16[0];\n#pragma unroll\n  for (int w = 1; w < 16; w++) total = total + lds16[w];\n  return total; // every thread returns the same value\n",
     "  asm volatile(\"\" ::\"v\"(s));\n  if (threadIdx.x == 0) g_sprof_t1 = wall_clock64();\n"
     "  s = butterfly64(s);\n  if ((threadIdx.x & 63u) == 0) lds16[threadIdx.x >> 6] = s;\n  __syncthreads();\n  double total = lds16[0];\n#pragma unroll\n  for (int w = 1; w < 16; w++) total = total + lds16[w];\n  return total; // every thread returns the same value\n"),
    ("  __shared__ double lds16[16];\n  // This launch sits on the critical path of every iteration: do not serialise the control block's\n",
     "  __shared__ double lds16[16];\n  const long long sp0 = wall_clock64();\n  // This launch sits on the critical path of every iteration: do not serialise the control block's\n"),
    ("  if (threadIdx.x == 0) cg_apply<MODE>(S, in, total, rr_hist, pAp_hist, defer_x);\n}\n\n// The same step on several ranks",
     "  const long long sp2 = wall_clock64();\n"
     "  if (threadIdx.x == 0) cg_apply<MODE>(S, in, total, rr_hist, pAp_hist, defer_x);\n"
     "  if (threadIdx.x == 0) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); const long long sp3 = wall_clock64(); const unsigned k = atomicAdd(&g_sprof_n, 1u) & 4095u;\n"
     "    g_sprof[4 * k] = sp0, g_sprof[4 * k + 1] = g_sprof_t1, g_sprof[4 * k + 2] = sp2, g_sprof[4 * k + 3] = sp3 | ((long long)MODE << 60); }\n"
     "}\n\n// The same step on several ranks"),
])
patch(W + "/sparsebench_amd/csrc/sbhip_matrix.inc.h", [
    ("uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }\n",
     "uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }\n"
     "extern \"C\" unsigned sb_lab_sprof(long long* out)\n{\n  HIP_CHECK(hipStreamSynchronize(g.stream));\n  unsigned n = 0;\n"
     "  HIP_CHECK(hipMemcpyFromSymbol(&n, HIP_SYMBOL(sbk::g_sprof_n), sizeof n));\n"
     "  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sbk::g_sprof), sizeof(long long) * 4096 * 4));\n  return n;\n}\n"),
])
os.makedirs(ROOT + "/labs", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-function",
                       "-shared", "-Wl,-soname,libsbhip.so", "-o", ROOT + "/labs/libsbhip_scalar.so", W + "/sparsebench_amd/csrc/sbhip.hip", "-ldl"])
print("built labs/libsbhip_scalar.so")
