"""bench.py's pre-flight: known answers before anything is timed.

Every rank solves PREFLIGHT_ITERS CG iterations on a 32^3-per-rank problem and on the bench's own bricks, with every SpMV kernel
that will be timed, on the data plane that is selected right now; the histories are checked against closed forms (r.r of the
prologue, p.Ap of the first body: exact integers at any size and rank count, sparsebench_amd/knownanswers.py), against the committed
oracle histories in the GPU's dot order (tests/golden/cg_hist_tree.json: bit for bit) and against every other rank's history.
check_history is pure (tests/test_known_answers.py runs it on the CPU)."""
import hashlib
import json
import os
import sys

from .context import quiet_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PREFLIGHT_ITERS = 20   # itermax of the pre-flight solves (and of the committed P-rank goldens)
PREFLIGHT_SMALL = 32   # brick edge of the small pre-flight problem


def golden_key(n, P, fmt, Cc, sigma):
    return "hpcg%d_x%d_%s" % (n, P, "crs" if fmt == "crs" else "scs_C%d_sigma%d" % (Cc, sigma))


def load_goldens():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "cg_hist_tree.json")))
    except (OSError, ValueError):
        return {}


def check_history(label, rr, pap, n, world, key, goldens):
    """One pre-flight solve against what is known about it.  Returns (record, problems)."""
    import numpy as np
    from sparsebench_amd import knownanswers as ka
    rec = {"case": label, "iterations": int(len(pap))}
    bad = []
    want_rr0, want_pap1 = float(ka.hpcg_rr0(n, n, n * world)), float(ka.hpcg_pAp1(n, n, n * world))
    if len(rr) < 2 or len(pap) < 1:
        return rec, ["%s: the solve produced no history (%d r.r, %d p.Ap values)" % (label, len(rr), len(pap))]
    rec["rr0"], rec["rr0_closed_form"] = float(rr[0]), want_rr0
    rec["pAp1"], rec["pAp1_closed_form"] = float(pap[0]), want_pap1
    if rr[0] != want_rr0:
        bad.append("%s: r.r of the prologue is %.17g, closed form %.17g" % (label, rr[0], want_rr0))
    if pap[0] != want_pap1:
        bad.append("%s: p.Ap of the first body is %.17g, closed form %.17g (first product that needs the halo)" % (label, pap[0], want_pap1))
    if not (np.all(np.isfinite(rr)) and np.all(np.isfinite(pap)) and np.all(pap > 0) and np.all(rr > 0)):
        bad.append("%s: the history holds non-finite or non-positive values" % label)
    g = goldens.get(key)
    rec["golden"] = key if g else None
    if g:
        grr = np.array([float(v) for v in g["rr"]])
        gpa = np.array([float(v) for v in g["pAp"]])
        m, q = min(len(grr), len(rr)), min(len(gpa), len(pap))
        if m < PREFLIGHT_ITERS - 2 or q < PREFLIGHT_ITERS - 2:
            bad.append("%s: only %d / %d values to compare with the golden history" % (label, m, q))
        elif not (np.array_equal(rr[:m], grr[:m]) and np.array_equal(pap[:q], gpa[:q])):
            d = np.nonzero(rr[:m] != grr[:m])[0]
            e = np.nonzero(pap[:q] != gpa[:q])[0]
            bad.append("%s: history differs from tests/golden/cg_hist_tree.json[%s]: first r.r mismatch at %s, first p.Ap "
                       "mismatch at %s" % (label, key, d[0] if len(d) else None, e[0] if len(e) else None))
        rec["golden_values_compared"] = int(m + q)
    return rec, bad


def preflight(ctx, plane_name, prob_full, n, modes, goldens):
    """PREFLIGHT_ITERS CG iterations on a 32^3-per-rank problem and on the bench's own bricks, once per SpMV kernel mode that will
    be timed, on the data plane that is selected right now.  Collective; returns (records, problems) identical on every rank."""
    from sparsebench_amd import hostapi
    a, rank, world = ctx.args, ctx.rank, ctx.world
    records, problems = [], []
    cases = [("32^3 per rank", PREFLIGHT_SMALL, None)]
    if n != PREFLIGHT_SMALL:
        cases.append(("bench bricks (%d^3 per rank)" % n, n, prob_full))
    else:
        cases = [("bench bricks (32^3 per rank)", n, prob_full)]
    for what, nn, pr in cases:
        own = pr is None
        if own:
            with quiet_stdout():
                pr = hostapi.Problem("generate", nn, nn, nn, fmt=a.fmt, Cc=a.C, sigma=a.sigma, rank=rank, size=world)
        lib_default = pr.pack_info()["mode"]
        seen = set()
        for mode in modes:
            # (the small problem has a default of its own: `None` among the modes stands for "the library's choice")
            got = pr.use_packed(lib_default if mode is None else mode)
            if got in seen:
                continue
            seen.add(got)
            label = "%s: %s, SpMV kernel mode %d" % (plane_name, what, got)
            cg = ctx.new_cg(pr, graph=False)
            cg.solve(PREFLIGHT_ITERS, 0.0)
            rr, pap = cg.history()
            cg.free()
            rec, bad = check_history(label, rr, pap, nn, world, golden_key(nn, world, a.fmt, a.C, a.sigma), goldens)
            rec["spmv_kernel_mode"] = got
            digest = hashlib.sha256(rr.tobytes() + pap.tobytes()).hexdigest()[:16]
            everyone = ctx.gather((digest, bad))
            digests = [d for d, _ in everyone]
            rec["history_sha256_by_rank"] = digests
            if len(set(digests)) != 1:
                bad = bad + ["%s: the ranks hold DIFFERENT histories (all-reduced scalars must be identical bits): %s" % (label, digests)]
            for r, (_, b) in enumerate(everyone):  # a problem seen by any rank is everybody's problem
                for msg in b:
                    if msg not in bad:
                        bad.append("rank %d: %s" % (r, msg))
            rec["ok"] = not bad
            records.append(rec)
            problems += bad
        if own:
            pr.free()
    return records, problems


def fail_preflight(ctx, records, problems, workload):
    """no rate from a run whose values are wrong: rank 0 prints a line with "value": null, every rank leaves with code 4"""
    if ctx.rank == 0:
        for msg in problems:
            sys.stderr.write("bench: PRE-FLIGHT FAILED: %s\n" % msg)
        print(json.dumps({"metric": "cg_iterations_per_s", "value": None, "ok": False, "unit": "iterations/s", "n_gpus": ctx.world,
                          "steps": ctx.K, "warmup": ctx.W, "error": "pre-flight check failed: nothing was timed",
                          "config": {"workload": workload},
                          "preflight": {"ok": False, "problems": problems, "checks": records}}), flush=True)
    if ctx.world > 1:
        ctx.L.sb_sync()
        ctx.dist.barrier()
    # (no sb_comm_finalize / destroy_process_group: the run is invalid, leave at once with the failure code)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(4)
