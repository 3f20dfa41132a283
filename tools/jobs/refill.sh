#!/bin/bash
out=$1; mkdir -p "$out"
b() { python3 bench.py --refill-min $1 --steps 40 --warmup 8 --no-cpu-baseline --no-legs "${@:2}" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('refill_min', sys.argv[1], ' '.join(sys.argv[2:]), '->', round(d['value']), 'Mpx/s  march', round(d['roofline']['kernel_ms'],4), 'draw', round(d['roofline']['draw_ms'],4))" "$@" | tee -a "$out/refill.txt"; }
for round in 1 2; do
  for r in 1 56 64; do
    b $r --frames-in-flight 4
    b $r --frames-in-flight 1 --scene g8 --max-iter 128
    b $r --frames-in-flight 1 --scene g64 --width 3840 --height 2160 --max-iter 512
    b $r --frames-in-flight 1 --scene g32s --width 3840 --height 2160
    b $r --frames-in-flight 1 --specialize 0
  done
done
