#!/usr/bin/env bash
# Same-box A/B of two builds of libsbhip.so inside CG (rocprofv3 kernel averages): the product library
# against labs/old/libsbhip.so, e.g. built from an earlier commit:
#   git archive <commit> | tar -x -C /tmp/oldsrc
#   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -shared \
#         -o labs/old/libsbhip.so /tmp/oldsrc/sparsebench_amd/csrc/sbhip.hip -ldl
# (an older library must export every symbol sparsebench_amd/capi.py lists; add stubs if needed).
# LD_PRELOAD makes the host library bind to the same old build that SBHIP_LIBRARY hands to ctypes.
# Run on the GPU box:  gpurun -- bash tools/ab_old_new.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in new old new old; do
  if [ $v = old ]; then export LD_PRELOAD=$PWD/labs/old/libsbhip.so SBHIP_LIBRARY=$PWD/labs/old/libsbhip.so; else unset LD_PRELOAD SBHIP_LIBRARY; fi
  rm -rf gpurun_out/ab_$v
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -o r1 -- python3 bench.py --no-cpu --steps 240 > gpurun_out/ab_$v.log 2>&1
  unset LD_PRELOAD SBHIP_LIBRARY
  python3 - <<PY
import csv
rows = {r["Name"].split("(")[0][-34:]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("gpurun_out/ab_$v/r1_kernel_stats.csv"))}
print("$v", "  ".join("%s=%.2f" % (k, v) for k, v in rows.items() if "cg_scalar_k<1" in k or "cg_scalar_k<2" in k or "update_p" in k or "dot_spans_k<3" in k or "scs64_pat<true" in k))
PY
done
