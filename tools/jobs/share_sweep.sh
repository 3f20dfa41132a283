#!/bin/bash
out=$1; mkdir -p "$out"
for f in 4 8; do for w in 0 8; do
  timeout -k 10 200 python tools/rank_share.py --worlds 4,8,16 --frames-in-flight $f --waves-per-tile $w 2>/dev/null | grep -v "rank [1-9]" >> "$out/sweep.txt"
done; done
timeout -k 10 200 python tools/rank_share.py --worlds 1 --frames-in-flight 8 2>/dev/null >> "$out/sweep.txt"
timeout -k 10 200 python tools/rank_share.py --worlds 1 --waves-per-tile 8 2>/dev/null >> "$out/sweep.txt"
cat "$out/sweep.txt"
