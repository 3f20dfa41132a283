#!/bin/bash
# GPU-box job (round 3): small launches without the sort kernel -- the whole GPU suite first (small images take the new path), then one rank's share.
out=$1; mkdir -p "$out"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$out/gpu_suite.log" 2>&1; echo "suite rc=$?" > "$out/status.txt"
tail -3 "$out/gpu_suite.log"
grep -q "suite rc=0" "$out/status.txt" || { tail -60 "$out/gpu_suite.log"; exit 1; }
for round in 1 2; do
echo "== raster order for small launches (default)" | tee -a "$out/share.txt"; timeout -k 10 300 python3 tools/rank_share.py --worlds 4,8 2>>"$out/err.log" | tee -a "$out/share.txt"
echo "== RM_SORT_SMALL=1 (the sort kernel for every launch)" | tee -a "$out/share.txt"; RM_SORT_SMALL=1 timeout -k 10 300 python3 tools/rank_share.py --worlds 4,8 2>>"$out/err.log" | tee -a "$out/share.txt"
done
python3 tools/time_kernel.py --scene g32 2>>"$out/err.log" | head -1 | tee -a "$out/share.txt"
cat "$out/status.txt"
