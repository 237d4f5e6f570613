"""The pieces of bench.py's JSON line that do not depend on how the numbers were obtained: the roofline block, the
PMC traffic look-up, byte counts of the CG loop.

Convention: `value`, `ms_per_step`, `phases_us` and `roofline` describe the loop whose SpMV streams the
REFERENCE's own Sell-C-sigma / CRS arrays (spmv_scs64 / spmv_crs_split): roofline.achieved = SURVEY 8d's algorithmic bytes per
launch (sb_matrix_spmv_bytes: the true-nnz formula) / the kernel's event time; cg_frac_of_roofline = iterations/s x the reference's
unfused op-list bytes (96 B/row + the SpMV's) / 8 TB/s.  The loop on the lossless compressed mirror -- which exploits the matrix's
repeating row shapes and runs out of the Infinity Cache, so that its time is not a statement about HBM -- is reported whole in the
`structure_exploiting` block."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def pmc_traffic(workload, kernel, version=None):
    """HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, as the guide prescribes) of `kernel` on
    `workload`, from the newest committed profiles/*_pmc_traffic.json whose entry was collected with
    THIS kernel source (content hash of sparsebench_amd/csrc/*, sparsebench_amd/srchash.py -- a kernel
    edit invalidates the entry whether or not anybody bumped a version string).  Returns (bytes, source,
    note): bytes is None -- never a stale constant -- when no pass matches, and the note says what is missing."""
    from sparsebench_amd import srchash
    version = srchash.csrc_hash()
    pdir = os.path.join(ROOT, "profiles")
    names = sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json")), reverse=True) \
        if os.path.isdir(pdir) else []
    stale = None
    for name in names:
        try:
            doc = json.load(open(os.path.join(pdir, name)))
        except (OSError, ValueError):
            continue
        e = doc.get(workload, {}).get(kernel)
        if not e:
            continue
        if e.get("source_hash", doc.get("source_hash")) == version:
            return e["bytes_per_launch"], "profiles/" + name, None
        stale = stale or "profiles/%s holds %s/%s for kernel source %r, the library was built from %r" % (
            name, workload, kernel, e.get("source_hash", doc.get("source_hash")), version)
    note = stale or "no committed PMC pass for %s / %s" % (workload, kernel)
    sys.stderr.write("bench: roofline.traffic = null: %s\n" % note)
    return None, None, note


def kernel_name(fmt, mode, crs_split=True):
    native = ("spmv_crs_split" if crs_split else "spmv_crs_stream") if fmt == "crs" else "spmv_scs64"
    return {0: native, 1: "spmv_scs64_packed", 2: "spmv_scs64_lds", 3: "spmv_scs64_pat", 5: "spmv_scs64_pat_masked"}[mode]


def roofline_block(kernel, nbytes, alg, us, launches, traffic, traffic_src, traffic_note, on_moved_bytes=False):
    """achieved = nbytes / average launch duration (HIP events on the layer's stream); frac = achieved / 8 TB/s.  For the contract's
    `roofline` nbytes IS SURVEY 8d's algorithmic figure (alg); the structure-exploiting block passes the bytes its kernel really
    moves and says so (on_moved_bytes)."""
    gbs = nbytes / (us * 1e-6) / 1e9 if launches and us > 0 else 0.0
    blk = {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
           "bytes_per_launch": nbytes, "algorithmic_bytes_per_launch": alg,
           "bytes_are": "moved by this kernel (less than the algorithmic figure)" if on_moved_bytes else
                        "SURVEY 8d algorithmic bytes (true-nnz formula, sb_matrix_spmv_bytes)",
           "avg_launch_us": us, "launches_timed": launches}
    if traffic:
        blk["traffic_over_bytes"] = traffic / nbytes
        if launches and us > 0:  # the same fraction on the bytes the PMC counters saw (gathers that miss the caches included)
            blk["frac_on_traffic"] = traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
    if traffic_note:
        blk["traffic_note"] = traffic_note
    return blk


def vector_bytes(nr, vector_phase=False):
    """bytes the fused loop's vector kernels move per iteration.  Separate launches: p update (+ the x update
    owed by the previous body) 40 B/row, r update + r.r partials 24 B/row.  One-launch vector phase: r, Ap, p, x
    read and r, p, x written once: 56 B/row.  Plus the partials written and read back."""
    return (56.0 if vector_phase else 64.0) * nr + 2 * 8.0 * (nr / 256.0)


def phase_table(ph):
    return {k: round(v[0], 3) for k, v in ph.items()} if ph else None


PARITY = {
    "checked_in_this_run": "pre-flight histories: closed forms exact, committed oracle histories bit for bit, all ranks identical, "
                           "every SpMV kernel that is timed",
    "bit_identical_to": "the pinned CPU oracle under the GPU's fixed dot order (tests/golden/cg_hist_tree.json; -m gpu tests at 64^3 / 128^3)",
    "vs_cg_with_exactly_rounded_dots": "<= 1e-12 relative per iteration at 128^3 (observed 2.2e-14; tests/golden/cg_hist_exact.json)",
    "vs_reference_cpu_history": "<= 1e-12 on 8^3..32^3 and on the irregular stand-in; at 64^3 / 128^3 bounded at 5e-11 / 6.5e-10: the "
                                "reference's own sequential ddot is 2.5e-11 / 3.2e-10 away from the exactly rounded history (its "
                                "summation error grows with n; no parallel order can follow it) -- north_star's 1e-12 is met against "
                                "the exact history at the benchmark size, not against the reference's rounding"}
