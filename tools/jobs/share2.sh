#!/bin/bash
out=$1; mkdir -p "$out"
for w in 0 8; do
  timeout -k 10 300 python tools/rank_share.py --worlds 1,2,4,8,16 --waves-per-tile $w 2>/dev/null | grep -v "rank [1-9]" >> "$out/share.txt"
done
timeout -k 10 200 python tools/rank_share.py --worlds 1,8 --width 3840 --height 2160 2>/dev/null | grep -v "rank [1-9]" >> "$out/share.txt"
timeout -k 10 200 python tools/rank_share.py --worlds 1,8 --gather 2>/dev/null | grep -v "rank [1-9]" >> "$out/share.txt"
cat "$out/share.txt"
