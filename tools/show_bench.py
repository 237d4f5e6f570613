"""show_bench.py <file> -- the few numbers of a bench.py JSON line one looks at first."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("it/s %.0f  ms/step %.5f  %s" % (d["value"], d["ms_per_step"], d.get("compression")))
r = d["roofline"]
print("%s  %.2f us  moved %.1f MB  %.0f GB/s moved  %.0f GB/s algorithmic" % (
    r["kernel"], r["avg_launch_us"], r["moved_bytes_per_launch"] / 1e6, r["moved_GBs"], r["achieved"]))
