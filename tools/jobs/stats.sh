#!/bin/bash
out=$1; mkdir -p "$out"
for m in ${MODES:-1 2 3 4}; do
  for sc in "g32 16 1920 1080 256" "g64 32 3840 2160 512"; do set -- $sc
    echo "== $1 mode $m" >> "$out/stats.txt"
    RM_JIT_PRUNE_STATS=$m python3 tools/wave_stats.py --scene $1 --leaves $2 --width $3 --height $4 --max-iter $5 --prune --balance 3 2>>"$out/err.log" | grep -E "lane occupancy|per iteration|iterations:|kernel span" >> "$out/stats.txt"
  done
done
cat "$out/stats.txt"
