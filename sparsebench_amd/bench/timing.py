"""bench.py's timed passes on one resident matrix.

A "step" is one CG iteration (loop body of solveCG, src/CGSolver.c:107-129).  Exactly K steps are timed between barrier + sync
pairs, in segments restarted from x0 = 0 outside the clock (r.r stays far from underflow); the max over ranks is taken.  Passes:
clean (no events: the rates), events (HIP events around every SpMV launch, on the layer's stream: the roofline legs), phases (an
event after every launch: the per-kernel breakdown), sustained (the clean loop over thousands of steps in one go)."""
import time

SEGMENT = 120  # iterations per timed segment


def measure(ctx, prob, modes, clean_all=True, phases=True, sustained=False):
    """modes: SpMV kernel modes (sb_matrix_use_packed) to time; the first is the one `value` is quoted on.  Every mode gets a clean
    pass when clean_all (else only the first).  Returns {mode: record}."""
    a, L, K, W = ctx.args, ctx.L, ctx.K, ctx.W
    prob.use_packed(modes[0])
    cg = ctx.new_cg(prob)

    def timed_pass(with_spmv_events, with_phases=False, steps=None):
        """exactly K loop bodies (`steps`: the sustained leg's count), in segments restarted from x0 = 0 outside the clock"""
        total, left, spmv_ms, spmv_n = 0.0, (steps or K), 0.0, 0
        phase_acc = {}
        while left > 0:
            seg = min(left, SEGMENT)
            cg.spmv_timing(False)
            cg.phase_timing(False)
            cg.start(itermax=W + 2 + seg, eps=0.0)  # prologue
            cg.run_iters(W + 1)                     # warm-up bodies, untimed
            before = cg.counters()
            cg.spmv_timing(with_spmv_events)
            cg.phase_timing(with_phases)
            ctx.barrier()
            t0 = time.perf_counter()
            cg.run_iters(seg)
            L.sb_sync()  # this rank's K steps are complete on its GPU ...
            dt = time.perf_counter() - t0
            ctx.barrier()  # ... and nobody moves on before all are (the max over ranks is taken below;
            #                   the gloo TCP barrier itself is control plane, not part of a CG step)
            after = cg.counters()
            if with_phases:
                for name, (us, cnt) in cg.phase_us().items():
                    acc = phase_acc.setdefault(name, [0.0, 0])
                    acc[0] += us * cnt
                    acc[1] += cnt
            cg.phase_timing(False)
            cg.finish()
            if after["stop"] and after["iters"] != W + 1 + seg:
                raise RuntimeError("bench: the loop exited early: %r" % after)
            if after["n_pAp"] - before["n_pAp"] != seg or after["iters"] != W + 1 + seg:
                raise RuntimeError("bench: the timed iterations did not all execute: %r -> %r" % (before, after))
            if with_spmv_events:
                ms, cnt = cg.spmv_ms()
                spmv_ms += ms
                spmv_n += cnt
            total += dt
            left -= seg
        return total, spmv_ms, spmv_n, {k: (v[0] / v[1], v[1] // max(1, steps or K)) for k, v in phase_acc.items() if v[1]}

    res = {}
    repeats = ctx.repeats
    for i, mode in enumerate(modes):
        got = prob.use_packed(mode)
        if got != mode or mode in res:
            continue
        t_clean = t_mine = None
        all_reps = []
        # the same loop over thousands of steps in one go (`sustained`), BEFORE the K-step windows: after idle time (the upload,
        # the host side of the pre-flight) the device runs the loop ~5 % slower for its first 50-150 ms under load, on fixed
        # memory, then settles (profiles/r04_placement_lab9.txt) -- a window of K = 20 steps is 3 ms of GPU work, so the windows
        # that follow a long run see the device as a solve of a few hundred iterations sees it
        t_sus = None
        if sustained and (i == 0 or clean_all) and a.sustained_steps > K:
            t_sus = ctx.rank_max(timed_pass(False, steps=a.sustained_steps)[0])
        if i == 0 or clean_all:
            mine, agreed = [], []
            for _ in range(repeats):
                dt = timed_pass(False)[0]
                mine.append(dt)
                agreed.append(ctx.rank_max(dt))  # the step ends when the slowest rank is done
            order = sorted(range(repeats), key=lambda j: agreed[j])
            mid = order[repeats // 2]        # the median repeat (the same one on every rank)
            t_clean, t_mine, all_reps = agreed[mid], mine[mid], agreed
        t_ev, ms, cnt = None, 0.0, 0
        if "events" in a.passes:
            t_ev, ms, cnt, _ = timed_pass(True)
        ph = timed_pass(False, True)[3] if phases and "phases" in a.passes and (i == 0 or clean_all) else None
        res[mode] = {"mode": mode, "fuse_p": cg.fuse_p(), "t_clean": t_clean, "t_mine": t_mine, "t_repeats": all_reps, "t_ev": t_ev,
                     "t_sus": t_sus, "spmv_us": 1e3 * ms / max(cnt, 1), "launches": cnt,
                     "launches_per_body": cg.launches_per_body(), "collectives_per_body": cg.collectives_per_body(),
                     "vector_phase": cg.vector_phase(),
                     "moved": prob.stream_bytes(), "alg": prob.spmv_bytes(), "phases": ph}
    prob.use_packed(modes[0])
    cg.free()
    return res
