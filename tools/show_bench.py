"""show_bench.py <file> -- the few numbers of a bench.py JSON line one looks at first (round-4 line layout: `value` / `roofline` on
the loop that streams the reference's arrays, the compressed-mirror loop in `structure_exploiting`)."""
import json, sys
d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
print("value %.0f it/s  ms/step %.5f  n_gpus %d  ok %s  kernel %s" % (d["value"], d["ms_per_step"], d["n_gpus"], d.get("ok"), d["config"].get("spmv_kernel")))
r = d.get("roofline")
if r:
    print("roofline: %-18s %7.2f us  %7.1f MB  %6.0f GB/s  frac %.3f  traffic %s" % (
        r["kernel"], r["avg_launch_us"], r["bytes_per_launch"] / 1e6, r["achieved"], r["frac"], r.get("traffic")))
    if r.get("device_stream_read_GBs"):
        print("          this device streams %.0f GB/s (plain read, 1 GiB): the kernel is at %.3f of that" % (
            r["device_stream_read_GBs"], r["achieved_over_device_stream_read"]))
for k in ("cg_frac_of_roofline", "cg_frac_of_hbm_peak_on_moved_bytes"):
    if d.get(k) is not None:
        print("%s = %.3f" % (k, d[k]))
if d.get("phases_us"):
    print("phases_us:", d["phases_us"])
if d.get("sustained"):
    print("sustained: %.0f it/s" % d["sustained"]["value"])
se = d.get("structure_exploiting")
if se:
    rm = se.get("roofline_on_moved_bytes") or {}
    print("structure_exploiting: %-16s %.0f it/s  ms/step %.5f  moved %.1f MB  speedup %.2f  launch %.2f us  frac on moved bytes %.3f  sustained %s" % (
        se["kernel"], se["value"], se["ms_per_step"], se["moved_bytes_per_launch"] / 1e6, se["algorithmic_speedup"],
        rm.get("avg_launch_us", 0.0), rm.get("frac_of_hbm_peak_on_moved_bytes", 0.0), (se.get("sustained") or {}).get("value")))
    print("   phases_us:", se.get("phases_us"))
for name, f in (d.get("formats") or {}).items():
    r = f["roofline"]
    print("  %-20s %6.0f it/s  %-18s %7.1f us  frac %.3f  fill %.3f  dot pass %s" % (
        name, f["cg_iterations_per_s"], r["kernel"], r["avg_launch_us"], r["frac"], f["fill"], f.get("separate_dot_pass")))
cb = d.get("cpu_baseline")
if cb:
    print("cpu: %.1f it/s on %d cores (%s) mpi leg: %s" % (cb["value"], cb["cores"], cb["kind"], cb.get("mpi_openmp")))
