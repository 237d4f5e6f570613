"""show_bench.py <file> -- the few numbers of a bench.py JSON line one looks at first."""
import json, sys
d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
print("it/s %.0f  ms/step %.5f  n_gpus %d  %s" % (d["value"], d["ms_per_step"], d["n_gpus"], d.get("compression")))
for key in ("roofline", "roofline_reference_layout"):
    r = d.get(key)
    if r:
        print("%-26s %-18s %7.2f us  %7.1f MB  %6.0f GB/s  frac %.3f  traffic %s" % (
            key, r["kernel"], r["avg_launch_us"], r["bytes_per_launch"] / 1e6, r["achieved"], r["frac"], r.get("traffic")))
for k in ("algorithmic_speedup", "cg_moved_GBs_per_gpu", "cg_frac_of_hbm_peak"):
    if k in d:
        print("%s = %.3f" % (k, d[k]))
for name, f in (d.get("formats") or {}).items():
    r = f["roofline"]
    print("  %-20s %6.0f it/s  %-18s %7.1f us  frac %.3f  fill %.3f" % (name, f["cg_iterations_per_s"], r["kernel"], r["avg_launch_us"], r["frac"], f["fill"]))
cb = d.get("cpu_baseline")
if cb:
    print("cpu: %.1f it/s on %d cores (%s) mpi leg: %s" % (cb["value"], cb["cores"], cb["kind"], cb.get("mpi_openmp")))
