#!/usr/bin/env python3
"""make_prof_lab_fusep.py -- labs/libsbhip_prof_fusep.so: the product library with per-tile wall_clock64 stamps in
spmv_prog_fusep (the SpMV that takes the p update) and an entry point sb_lab_prof().  Scratch copy under /tmp; repository
files untouched.  Read with:  SBHIP_LIBRARY=$PWD/labs/libsbhip_prof_fusep.so python tools/make_prof_lab_fusep.py report [n] [sigma]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    W = "/tmp/sb_prof_fusep"
    shutil.rmtree(W, ignore_errors=True)
    os.makedirs(W + "/sparsebench_amd")
    shutil.copytree(ROOT + "/include", W + "/include")
    shutil.copytree(ROOT + "/sparsebench_amd/csrc", W + "/sparsebench_amd/csrc")

    def patch(path, pairs):
        s = open(path).read()
        for old, new in pairs:
            if s.count(old) != 1:
                sys.exit("anchor not unique/present in %s:\n%s" % (path, old))
            s = s.replace(old, new)
        open(path, "w").write(s)

    T = "template <int CPT, bool SKIPPAD, bool HALO, bool MAPPED>\n__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(72))) void spmv_prog_fusep("
    patch(W + "/sparsebench_amd/csrc/pack.hip.h", [
        (T, "__device__ long long g_prof[8192 * 8];\n#define PROF(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_prof[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)\n" + T),
        ("  double* sq = lds;\n  double* sx = lds + 16;\n  constexpr int CW   = CPT / 4;\n  constexpr int LONG = CPT == 8 ? 4 : 3;\n  constexpr int WB   = 3 * LONG + 3;\n  constexpr int H1   = 8;",
         "  PROF(0);\n  double* sq = lds;\n  double* sx = lds + 16;\n  constexpr int CW   = CPT / 4;\n  constexpr int LONG = CPT == 8 ? 4 : 3;\n  constexpr int WB   = 3 * LONG + 3;\n  constexpr int H1   = 8;"),
        ("  const uint32_t tile = field(46), flags = field(31), win = field(3);\n  if (HALO &&", "  const uint32_t tile = field(46), flags = field(31), win = field(3);\n  PROF(1);\n  if (HALO &&"),
        ("  { // second half: the same registers again", "  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  PROF(2);\n  { // second half: the same registers again"),
        ("  if (MAPPED && threadIdx.x == 0) sx[0] = r[padCol] + beta * pold[padCol]; // (behind this thread's own store to slot 0)\n  __syncthreads();\n",
         "  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  PROF(3);\n  if (MAPPED && threadIdx.x == 0) sx[0] = r[padCol] + beta * pold[padCol]; // (behind this thread's own store to slot 0)\n  __syncthreads();\n  PROF(4);\n"),
        ("    pl0[c] = butterfly64(t2);\n", "    pl0[c] = butterfly64(t2);\n    PROF(5 + c);\n"),
    ])
    patch(W + "/sparsebench_amd/csrc/sbhip_matrix.inc.h", [
        ("uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }\n",
         "uint32_t sb_matrix_pattern_classes(const sb_matrix* m) { return pat_of(m)->nPatClasses; }\n"
         "extern \"C\" void sb_lab_prof(long long* out)\n{\n  HIP_CHECK(hipStreamSynchronize(g.stream));\n"
         "  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sbk::g_prof), sizeof(long long) * 8192 * 8));\n}\n"),
    ])
    os.makedirs(ROOT + "/labs", exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-function",
                           "-shared", "-Wl,-soname,libsbhip.so", "-o", ROOT + "/labs/libsbhip_prof_fusep.so", W + "/sparsebench_amd/csrc/sbhip.hip", "-ldl"])
    print("built labs/libsbhip_prof_fusep.so")


def report(n, sigma):
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    from sparsebench_amd import capi, hostapi
    L = capi.init(0)
    p = hostapi.Problem("generate", n, n, n, fmt="scs", Cc=64, sigma=sigma)
    cg = hostapi.CG(p, fuse_p=1)
    assert cg.fuse_p() == 1
    cg.solve(40, 0.0)
    prof = np.zeros(8192 * 8, dtype=np.int64)
    L.sb_lab_prof(prof.ctypes.data_as(C.c_void_p))
    nT = min(8192, (p.nChunks + 7) // 8)
    q = prof.reshape(8192, 8)[:nT, :7].astype(np.float64) / 100.0
    t0 = q[:, 0].min()
    print("spmv_prog_fusep, HPCG %d^3 sigma %d, %d tiles; last launch of a 40-iteration solve; kernel span %.2f us" % (n, sigma, nT, q[:, 6].max() - t0))
    names = ["start -> header decoded", "-> first half's p, r back (exit test passed)", "-> second half's p, r back", "-> barrier passed",
             "-> chunk 0 summed (own-row loads, programs, stores)", "-> chunk 1 summed"]
    for i, nm in enumerate(names):
        d = q[:, i + 1] - q[:, i]
        print("  %-52s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f us" % (nm, d.mean(), *np.percentile(d, [10, 50, 90])))
    first = (q[:, 0] - t0) < 3.0  # tiles of the first round (dispatched while the chip is still filling) against the rest
    for label, sel in (("first round (started < 3 us)", first), ("later tiles", ~first)):
        if sel.sum():
            print("  %s: %d tiles; " % (label, int(sel.sum())) + ", ".join("%s %.2f" % (nm.split("(")[0].replace("->", "").strip()[:24], (q[sel, i + 1] - q[sel, i]).mean())
                                                                        for i, nm in enumerate(names)) + "; life %.2f us" % (q[sel, 6] - q[sel, 0]).mean())
    life = q[:, 6] - q[:, 0]
    print("  tile life                                            mean %.2f  p50 %.2f  p90 %.2f us" % (life.mean(), *np.percentile(life, [50, 90])))
    print("  start times: p25 %.2f p50 %.2f p75 %.2f p100 %.2f us" % tuple(np.percentile(np.sort(q[:, 0] - t0), [25, 50, 75, 100])))
    st, en = q[:, 0] - t0, q[:, 6] - t0
    print("  tiles in flight at t us:", ", ".join("%g: %d" % (t, int(((st <= t) & (en > t)).sum())) for t in (0.5, 1, 2, 3, 5, 8, 12, 16, 20, 25)))
    print("  tiles started by t us:  ", ", ".join("%g: %d" % (t, int((st <= t).sum())) for t in (0.2, 0.5, 1, 2, 3, 5, 8, 10, 12)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        report(int(sys.argv[2]) if len(sys.argv) > 2 else 128, int(sys.argv[3]) if len(sys.argv) > 3 else 256)
    else:
        build()
