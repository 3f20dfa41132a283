#!/bin/bash
# GPU-box job: duration of the pre-pass kernel for several sample-parallel thresholds (kernel trace)
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in ${NEEDS:-24 0 64}; do
  export RM_PRE_NEED_MAX=$n
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_$n" -- python3 bench.py --steps 30 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --no-legs > "$out/kt_$n.log" 2>&1
  python3 - "$out/kt_$n" "$n" <<'PY' | tee -a "$out/prepass.txt"
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "rm_tile_pre_v5" in r["Name"] or "rm_render_v5" in r["Name"]:
            print("need_max %s: %-20s calls %s  mean %.1f us  min %.1f us" % (sys.argv[2], r["Name"][:20], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  rm -rf "$out/kt_$n"
done
