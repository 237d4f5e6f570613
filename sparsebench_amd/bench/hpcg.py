"""bench.py, workload `hpcg` (BASELINE.json configs[2] / [3]): HPCG 27-point stencil, n^3 rows per GPU, Sell-C-sigma (or CRS), one
rank per GPU, bricks stacked in z (weak scaling).

Two loops are timed on the one resident matrix, each validated by the pre-flight before it is timed:
  * the section-8d loop: the SpMV streams the reference's own arrays (kernel mode 0: spmv_scs64 / spmv_crs_split).  It carries
    `value`, `ms_per_step`, `phases_us`, `roofline` (algorithmic bytes / event time) and `cg_frac_of_roofline`;
  * the structure-exploiting loop: the SpMV runs on the lossless compressed mirror (masked row programs, p update inside the SpMV
    launch).  It is reported whole in `structure_exploiting` (its rate, the bytes it moves, its fraction of the HBM peak on MOVED
    bytes, algorithmic_speedup, sustained).
N > 1: the communicator's data plane (RCCL all-reduce + send/recv) is validated and timed FIRST and handed to the supervisor as a
provisional line (checkpoint); then the peer-mapped plane (`value`), then -- structure-exploiting kernel only -- the halo push inside
the SpMV launch.  A peer-mapped plane that fails its pre-flight degrades the run to the communicator's line (exit 0, "ok": false)."""
import ctypes
import json
import os
import sys

from .context import quiet_stdout
from .line import HBM_PEAK_GBS, PARITY, kernel_name, phase_table, pmc_traffic, roofline_block, vector_bytes
from .preflight import fail_preflight, load_goldens, preflight
from .supervisor import MARK
from .timing import measure

SE_NOTE = ("the same CG loop with the SpMV on the lossless compressed mirror (masked row programs over LDS x windows; where every chunk is a "
           "row program the p update rides inside the SpMV launch): it exploits the matrix's repeating row shapes, moves %.1f MB per SpMV "
           "launch instead of the reference layout's %.1f MB, and at this size its whole working set (mirror + five vectors) stays in the "
           "256 MiB Infinity Cache across iterations -- so its rate is a real, bit-identical solver rate but NOT an HBM-roofline figure "
           "(SURVEY 8d bytes / its time would exceed the HBM peak); frac_of_hbm_peak_on_moved_bytes is a latency statement")


def run(ctx, cpu):
    from sparsebench_amd import hostapi
    a, L, rank, world, K, W = ctx.args, ctx.L, ctx.rank, ctx.world, ctx.K, ctx.W
    n = a.n if a.n > 0 else 128
    with quiet_stdout():
        prob = hostapi.Problem("generate", n, n, n, fmt=a.fmt, Cc=a.C, sigma=a.sigma, rank=rank, size=world)
    lib_default = prob.use_packed(a.pack_mode) if a.pack_mode >= 0 else prob.pack_info()["mode"]
    if a.loops == "reference" or lib_default == 0:
        modes = [0]
    elif a.loops == "structure":
        modes = [lib_default]   # lab use: `value` is then the structure-exploiting loop and the line says so
    else:
        modes = [0, lib_default]
    primary = modes[0]
    se_mode = lib_default if lib_default != 0 and lib_default in modes else None
    workload = "hpcg_27pt_%d^3_per_gpu_%s_C%d_sigma%d" % (n, a.fmt, a.C, a.sigma)
    p2p_dots, p2p_halo = (L.sb_comm_p2p_enabled(), L.sb_halo_p2p_enabled(prob.halo)) if world > 1 else (0, 0)
    second_plane = world > 1 and (p2p_dots or p2p_halo) and not a.no_rccl_leg
    coll = "rccl" if a.transport == "rccl" else "host_staged_gloo"
    crs_split = bool(L.sb_matrix_crs_kernel(prob.matrix))
    goldens = load_goldens()

    def plane_name(plane):
        return "one GPU" if world == 1 else ("peer-mapped data plane" if plane and (p2p_dots or p2p_halo) else "%s data plane" % coll)

    def kern_of(rec):
        return "spmv_prog_fusep" if rec["fuse_p"] else kernel_name(a.fmt, rec["mode"], crs_split)

    def se_block(rec, plane_res=None, with_roofline=True):
        """the structure-exploiting loop, whole"""
        moved, alg = rec["moved"], rec["alg"]
        if rec["fuse_p"]:
            # the SpMV launch also takes the p update: + r and p_old read, p_new written, x read and written = 40 B/row,
            # the same 64 B/row of vector traffic per iteration as with the separate kernel
            moved, alg = moved + 40.0 * prob.nr, alg + 40.0 * prob.nr
        it = K / rec["t_clean"]
        cg_moved = moved + vector_bytes(prob.nr, rec["vector_phase"]) - (40.0 * prob.nr if rec["fuse_p"] else 0.0)
        blk = {"kernel": kern_of(rec), "spmv_kernel_mode": rec["mode"], "value": world * it, "ms_per_step": 1e3 * rec["t_clean"] / K,
               "ms_per_step_repeats": [1e3 * t / K for t in rec["t_repeats"]],
               "launches_per_iteration": rec["launches_per_body"], "collective_calls_per_iteration": rec["collectives_per_body"],
               "p_update_inside_spmv": bool(rec["fuse_p"]),
               "moved_bytes_per_launch": moved, "algorithmic_bytes_per_launch": alg, "algorithmic_speedup": alg / moved,
               "cg_moved_bytes_per_iteration": cg_moved,
               "cg_frac_of_hbm_peak_on_moved_bytes": cg_moved * it / 1e9 / HBM_PEAK_GBS,
               "phases_us": phase_table(rec["phases"]),
               "ms_per_step_with_events": (1e3 * rec["t_ev"] / K) if rec["t_ev"] else None,
               "note": SE_NOTE % (rec["moved"] / 1e6, rec["alg"] / 1e6)}
        if with_roofline:
            tr = pmc_traffic(workload, kern_of(rec)) if world == 1 else (None, None, "N > 1")
            r = roofline_block(kern_of(rec), moved, alg, rec["spmv_us"], rec["launches"], *tr, on_moved_bytes=True)
            r["frac_of_hbm_peak_on_moved_bytes"] = r.pop("frac")
            blk["roofline_on_moved_bytes"] = r
        if rec.get("t_sus"):
            ks = a.sustained_steps
            blk["sustained"] = {"steps": ks, "value": world * ks / rec["t_sus"], "ms_per_step": 1e3 * rec["t_sus"] / ks}
        return blk

    def compact_line(res, plane, recs, steps_ms):
        """a complete line of the contract quoted on ONE validated data plane (the provisional line handed to the supervisor behind
        the communicator-plane checkpoint; the degraded line when the peer-mapped plane fails its pre-flight)"""
        c = res[primary]
        it = K / c["t_clean"]
        line = {"metric": "cg_iterations_per_s", "value": world * it, "ok": True,
                "unit": "iterations/s (one iteration = one %d^3-brick CG step; summed over GPUs)" % n,
                "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * c["t_clean"] / K,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": workload, "rows_per_gpu": prob.nr, "nnz_per_gpu": prob.nnzTrue, "index_type": "u32",
                           "parallelism": "1d_block_row_x%d" % world, "data_plane": plane_name(plane),
                           "halo_exchange": "rccl_send_recv" if a.transport == "rccl" else "host_staged_gloo",
                           "dot_allreduce": coll, "spmv_kernel": kern_of(c), "launches_per_iteration": c["launches_per_body"],
                           "p_update_inside_spmv": bool(c["fuse_p"]), "library": ctx.version},
                "timed_repeats": ctx.repeats, "ms_per_step_repeats": [1e3 * t / K for t in c["t_repeats"]],
                "global_iterations_per_s": it, "phases_us": phase_table(c["phases"]),
                "roofline": roofline_block(kern_of(c), c["alg"], c["alg"], c["spmv_us"], c["launches"], None, None, "N > 1"),
                "cg_frac_of_roofline": it * (96.0 * prob.nr + c["alg"]) / 1e9 / HBM_PEAK_GBS,
                "per_rank": {"ms_per_step": steps_ms}, "preflight": {"ok": True, "checks": recs},
                "device": L.sb_device_name().decode(), "cpu_baseline": None}
        if se_mode is not None and se_mode in res and res[se_mode]["t_clean"]:
            line["structure_exploiting"] = se_block(res[se_mode], with_roofline=False)
        return line

    # pre-flight on every data plane that will be timed, before that plane is timed.  N > 1 with both planes up: the
    # communicator's plane FIRST (validated, timed, handed to the supervisor as a provisional line: the checkpoint), so
    # that a failure of the peer-mapped plane on its first contact with real links still leaves a validated rate.
    checks, coll_problems = [], []
    res_coll, degraded, provisional = None, None, None
    if second_plane:
        L.sb_comm_data_plane(0)
        if not a.no_preflight:
            checks, coll_problems = preflight(ctx, plane_name(0), prob, n, modes, goldens)
        if coll_problems:
            # the communicator's plane gives wrong results here: nothing is timed on it, no checkpoint; the run goes on to the
            # peer-mapped plane, whose own pre-flight decides whether there is a rate at all (both wrong: exit code 4)
            if rank == 0:
                for msg in coll_problems:
                    sys.stderr.write("bench: PRE-FLIGHT FAILED on the communicator's data plane: %s\n" % msg)
            L.sb_comm_data_plane(1)
        else:
            res_coll = measure(ctx, prob, modes)
            L.sb_comm_data_plane(1)
            coll_ms = ctx.gather(1e3 * res_coll[primary]["t_mine"] / K)
            provisional = compact_line(res_coll, 0, list(checks), coll_ms) if rank == 0 else None
            if ctx.supervised:
                if rank == 0:
                    print(MARK + "provisional " + json.dumps(provisional), flush=True)
                ctx.barrier()
                print(MARK + "checkpoint", flush=True)
            if os.environ.get("SB_BENCH_TEST_DIE_AFTER_CHECKPOINT") == str(rank):  # test hook: a crash in the peer-mapped legs
                sys.stderr.write("bench: rank %d: SB_BENCH_TEST_DIE_AFTER_CHECKPOINT is set, exiting with code 9 (test hook)\n" % rank)
                os._exit(9)
    if not a.no_preflight:
        recs, bad = preflight(ctx, plane_name(1), prob, n, modes, goldens)
        checks += recs
        if bad and coll_problems:
            fail_preflight(ctx, checks, coll_problems + bad, workload)
        if bad and second_plane:
            # the peer-mapped plane gives WRONG results here, the communicator's plane passed: no rate from the former, the
            # line is quoted on the latter and says so (exit code 0: a validated rate; the failure is in the line and on stderr)
            degraded = bad
        elif bad:
            fail_preflight(ctx, checks, bad, workload)
    if degraded:
        if rank == 0:
            for msg in degraded:
                sys.stderr.write("bench: PRE-FLIGHT FAILED on the peer-mapped data plane: %s\n" % msg)
            sys.stderr.write("bench: DEGRADED: the line is quoted on the communicator's data plane, which passed\n")
            provisional["ok"] = False
            provisional["degraded"] = {"why": "the peer-mapped data plane failed its pre-flight; nothing was timed on it",
                                       "value_is_quoted_on": provisional["config"]["data_plane"], "problems": degraded}
            provisional["preflight"] = {"ok": False, "checks": checks, "problems": degraded,
                                        "ok_on_the_plane_value_is_quoted_on": True}
            print(json.dumps(provisional), flush=True)
        L.sb_comm_data_plane(0)
        ctx.barrier()
        prob.free()
        return None

    res = measure(ctx, prob, modes, sustained=True)
    # third leg, peer-mapped halo and structure-exploiting kernel only: the halo push inside the SpMV launch (one launch fewer per
    # body).  Which variant is faster can only be decided with one rank per GPU, i.e. by this very run on a real node; validated
    # by its own pre-flight, and a failure here does not invalidate `value` (the variant is simply reported as failed).
    res_inside, inside_checks, inside_problems = None, [], []
    if world > 1 and p2p_halo and se_mode is not None and se_mode >= 3 and not a.no_push_inside_leg:
        L.sb_comm_halo_push_inside(1)
        if not a.no_preflight:
            inside_checks, inside_problems = preflight(ctx, "peer-mapped data plane, push inside the SpMV launch", prob, n, [se_mode], goldens)
        if not inside_problems:
            res_inside = measure(ctx, prob, [se_mode])
        L.sb_comm_halo_push_inside(0)
        prob.use_packed(primary)
    rccl = (ctypes.c_int * 3)()
    has_rccl = L.sb_comm_rccl_info(rccl) if world > 1 else 0

    def mine(r, m):
        return {"ms_per_step": 1e3 * r[m]["t_mine"] / K, "phases_us": phase_table(r[m]["phases"])} if r and m in r and r[m]["t_mine"] else None

    per_rank = ctx.gather({"rank": rank, "device": ctx.device, "spmv_mode_structure_exploiting": se_mode,
                           "rccl": list(rccl) if has_rccl else None,
                           "main": {m: mine(res, m) for m in modes}, "coll": {m: mine(res_coll, m) for m in modes},
                           "inside": mine(res_inside, se_mode)})
    out = None
    if rank == 0:
        d = res[primary]
        it_s = K / d["t_clean"]
        kern = kern_of(d)
        tr = pmc_traffic(workload, kern) if world == 1 else (None, None, "N > 1")
        roof = roofline_block(kern, d["alg"], d["alg"], d["spmv_us"], d["launches"], *tr)
        cg_alg = 96.0 * prob.nr + d["alg"]  # the reference's unfused op list on its own layout
        cg_moved = d["moved"] + vector_bytes(prob.nr, d["vector_phase"])
        steps_ms = [r["main"][primary]["ms_per_step"] for r in per_rank]
        if primary != 0:
            roof = roofline_block(kern, d["moved"] + (40.0 * prob.nr if d["fuse_p"] else 0.0), d["alg"], d["spmv_us"], d["launches"], *tr,
                                  on_moved_bytes=True)
            roof["note"] = "lab run (--loops structure): `value` and this block are the structure-exploiting loop, on MOVED bytes"
        else:
            roof["note"] = ("the SpMV streams the reference's own %s arrays (12 B per stored element): bytes = SURVEY 8d's algorithmic "
                            "figure, no use of the matrix's structure" % ("CRS" if a.fmt == "crs" else "Sell-C-sigma"))
        if world == 1 and roof["achieved"] > 0:
            # informational: what THIS device streams (a plain read of a fresh 1 GiB buffer, best of three allocations) next to the
            # 8 TB/s of the data sheet that `frac` is taken against -- measured after all timed passes
            stream = max(L.sb_debug_stream_read_gbs(1 << 30, 20) for _ in range(3))
            roof["device_stream_read_GBs"] = stream
            roof["achieved_over_device_stream_read"] = roof["achieved"] / stream
        out = {
            "metric": "cg_iterations_per_s",
            "value": world * it_s,
            "ok": True,
            "unit": "iterations/s (one iteration = one %d^3-brick CG step; summed over GPUs)" % n,
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * d["t_clean"] / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "rows_per_gpu": prob.nr, "nnz_per_gpu": prob.nnzTrue, "index_type": "u32",
                       "parallelism": "1d_block_row_x%d" % world,
                       "spmv_kernel": kern,
                       "spmv_layout": "the reference's own arrays (SURVEY 8d)" if primary == 0 else "compressed mirror (lab run)",
                       "transport": ("none" if world == 1 else "rccl_xgmi" if a.transport == "rccl"
                                     else "host_staged_gloo (rehearsal)"),
                       "halo_exchange": ("none" if world == 1 else "peer_mapped_push_pull" if p2p_halo
                                         else "rccl_send_recv" if a.transport == "rccl" else "host_staged_gloo"),
                       "dot_allreduce": ("none" if world == 1 else "in_kernel_peer_mapped" if p2p_dots
                                         else "rccl" if a.transport == "rccl" else "host_staged_gloo"),
                       "dot_allreduce_reason": (L.sb_comm_p2p_reason().decode() if world > 1 else None),
                       "halo_exchange_reason": (L.sb_halo_p2p_reason(prob.halo).decode() if world > 1 else None),
                       "rccl_ranks": (rccl[0] if has_rccl else None),
                       "spmv_x_staging": "l1_l2_gather (LDS staging measured neutral at 12 B/element)" if primary == 0 else "lds_window",
                       "fused_dots": True, "vector_phase_one_launch": bool(d["vector_phase"]),
                       "launches_per_iteration": d["launches_per_body"],
                       "collective_calls_per_iteration": d["collectives_per_body"],
                       "p_update_inside_spmv": bool(d["fuse_p"]),
                       "spmv_kernel_mode_structure_exploiting_by_rank": [r["spmv_mode_structure_exploiting"] for r in per_rank],
                       "device_by_rank": [r["device"] for r in per_rank],
                       "hip_graph": bool(a.graph), "library": ctx.version,
                       # which device memory the streamed arrays and the loop's vectors sit in is chosen by measurement at upload
                       # (DESIGN 4.1): what that tuner saw on this rank
                       "placement": prob.placement_report()},
            "timed_repeats": ctx.repeats,
            "ms_per_step_repeats": [1e3 * t / K for t in d["t_repeats"]],
            "global_iterations_per_s": it_s,
            "roofline": roof,
            "cg_reference_oplist_bytes_per_iteration": cg_alg,
            "cg_frac_of_roofline": (it_s * cg_alg / 1e9 / HBM_PEAK_GBS) if primary == 0 else None,
            "cg_roofline_iterations_per_s_per_gpu": HBM_PEAK_GBS * 1e9 / cg_alg,
            "cg_moved_bytes_per_iteration": cg_moved,
            "cg_frac_of_hbm_peak_on_moved_bytes": cg_moved * it_s / 1e9 / HBM_PEAK_GBS,
            "ms_per_step_with_events": (1e3 * d["t_ev"] / K) if d["t_ev"] else None,
            "phases_us": phase_table(d["phases"]),
            "preflight": ({"ok": not coll_problems, "checks": checks,
                           **({"problems": coll_problems, "ok_on_the_plane_value_is_quoted_on": True} if coll_problems else {})}
                          if not a.no_preflight else {"ok": None, "skipped": "--no-preflight"}),
            "compression": prob.pack_info(),
            "device": L.sb_device_name().decode(),
            "parity": PARITY,
        }
        if d.get("t_sus"):
            ks = a.sustained_steps
            out["sustained"] = {"steps": ks, "value": world * ks / d["t_sus"], "ms_per_step": 1e3 * d["t_sus"] / ks,
                                "note": "the same loop, clean, over %d steps in one go, run BEFORE the K-step windows (informational; `value` is "
                                        "the K = %d figure the contract asks for): after idle time the device runs the loop ~5 %% slower "
                                        "for its first 50-150 ms under load, on fixed memory (profiles/r04_placement_lab9.txt); this leg "
                                        "absorbs that, the windows after it see the device as a solve of hundreds of iterations does" % (ks, K)}
        if se_mode is not None and se_mode in res and se_mode != primary:
            se = se_block(res[se_mode])
            out["structure_exploiting"] = se
            out["cg_iterations_per_s_by_spmv_kernel"] = {
                kern + " (streams the reference's arrays: SURVEY 8d bytes, no use of structure) = value": world * it_s,
                se["kernel"] + " (lossless compressed mirror: exploits the matrix's repeating row shapes)": se["value"]}
        if world > 1:
            out["per_rank"] = {"ms_per_step": steps_ms, "ms_per_step_min": min(steps_ms), "ms_per_step_max": max(steps_ms),
                               "device": [r["device"] for r in per_rank], "rccl": [r["rccl"] for r in per_rank],
                               "phases_us": [r["main"][primary]["phases_us"] for r in per_rank]}
            ph_all = [p for p in out["per_rank"]["phases_us"] if p]
            if ph_all:
                out["phases_us_max_over_ranks"] = {k: max(p.get(k, 0.0) for p in ph_all) for k in ph_all[0]}
            if "structure_exploiting" in out:
                se_rank = [r["main"][se_mode] for r in per_rank]
                out["structure_exploiting"]["per_rank_ms_per_step"] = [r["ms_per_step"] if r else None for r in se_rank]
                out["structure_exploiting"]["phases_us_by_rank"] = [r["phases_us"] if r else None for r in se_rank]
        if res_coll:
            c = res_coll[primary]
            out["rccl_only"] = {
                "value": world * K / c["t_clean"], "ms_per_step": 1e3 * c["t_clean"] / K,
                "ms_per_step_repeats": [1e3 * t / K for t in c["t_repeats"]],
                "halo_exchange": "rccl_send_recv" if a.transport == "rccl" else "host_staged_gloo",
                "dot_allreduce": coll, "spmv_kernel": kern_of(c), "launches_per_iteration": c["launches_per_body"],
                "collective_calls_per_iteration": c["collectives_per_body"],
                "per_rank_ms_per_step": [r["coll"][primary]["ms_per_step"] for r in per_rank], "phases_us": phase_table(c["phases"]),
                "phases_us_by_rank": [r["coll"][primary]["phases_us"] for r in per_rank],
                "note": "same bricks, same K steps, peer-mapped paths switched off (sb_comm_data_plane(0)): the communicator's "
                        "all-reduce and send/recv carry the dots and the halo"}
            if se_mode is not None and se_mode in res_coll and se_mode != primary:
                s = res_coll[se_mode]
                out["rccl_only"]["structure_exploiting"] = {
                    "value": world * K / s["t_clean"], "ms_per_step": 1e3 * s["t_clean"] / K, "kernel": kern_of(s),
                    "launches_per_iteration": s["launches_per_body"], "collective_calls_per_iteration": s["collectives_per_body"],
                    "phases_us": phase_table(s["phases"])}
        if res_inside:
            c = res_inside[se_mode]
            out["structure_exploiting"]["push_inside"] = {
                "value": world * K / c["t_clean"], "ms_per_step": 1e3 * c["t_clean"] / K,
                "ms_per_step_repeats": [1e3 * t / K for t in c["t_repeats"]], "launches_per_iteration": c["launches_per_body"],
                "per_rank_ms_per_step": [r["inside"]["ms_per_step"] for r in per_rank], "phases_us": phase_table(c["phases"]),
                "phases_us_by_rank": [r["inside"]["phases_us"] for r in per_rank],
                "preflight": {"ok": True, "checks": inside_checks},
                "note": "the same K steps with the rank's halo push carried by the first workgroups of the SpMV launch "
                        "(sb_comm_halo_push_inside(1)) instead of a push launch of its own; not the default -- ranks sharing a "
                        "GPU (rehearsals) keep each other's pushes off the CUs, so only a run with one rank per GPU can rank the two"}
        elif inside_problems:
            out["structure_exploiting"]["push_inside"] = {"value": None, "preflight": {"ok": False, "problems": inside_problems,
                                                                                       "checks": inside_checks}}
        if coll_problems:
            out["ok"] = False
            out["rccl_only"] = {"value": None, "preflight": {"ok": False, "problems": coll_problems},
                                "note": "the communicator's data plane failed its pre-flight: nothing was timed on it; `value` is the "
                                        "peer-mapped plane's, which passed"}
            out["degraded"] = {"why": "the communicator's data plane failed its pre-flight", "value_is_quoted_on": plane_name(1),
                               "problems": coll_problems}
        elif res_coll is None and world > 1:
            out["rccl_only"] = {"note": "not timed separately: " + (
                "--no-rccl-leg" if a.no_rccl_leg else "the peer-mapped paths are off, `value` IS the communicator's data plane")}
        out["cpu_baseline"] = cpu
    prob.free()
    return out
