#!/bin/bash
# GPU-box job: how much the dispatch order of the tiles matters (RM_OPT_BALANCE 0 arrival / 1 pending pixels / 3 last frame's durations)
out=$1; mkdir -p "$out"
for round in 1 2; do
for sc in "g8 128" "g32 256"; do set -- $sc
for bal in 3 1 0; do
  timeout -k 10 200 python bench.py --scene $1 --max-iter $2 --steps 40 --warmup 10 --frames-in-flight 1 --no-cpu-baseline --no-legs --balance $bal > "$out/b.json" 2>> "$out/err.log"
  python3 -c "
import json; d=json.load(open('$out/b.json')); print('$1 balance $bal: march %.4f ms  draw %.4f ms' % (d['roofline']['kernel_ms'], d['roofline']['draw_ms']))" | tee -a "$out/balance.txt"
done; done; done
