#!/bin/bash
# GPU-box job: kernel trace of a small rank share (N = 16: 80 rows) to see what the per-draw floor is made of.
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 tools/rank_share.py --worlds ${WORLDS:-16} --steps 60 > "$out/kt.log" 2>&1
cat "$out"/kt/*/*_kernel_stats.csv > "$out/kernel_stats.csv"
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-48:]
t0 = int(tail[0]["Start_Timestamp"])
with open(out + "/timeline.txt", "w") as o:
    for r in tail:
        o.write("%-28s queue %s  start %8.1f us  dur %7.1f us  grid %s wg %s\n" % (r["Kernel_Name"][:28], r.get("Queue_Id", "?"),
                (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
print(open(out + "/timeline.txt").read())
PY
rm -rf "$out/kt"
