#!/bin/bash
# GPU-box job (round 3): does wave-level culling pay for 4-leaf programs now (BASELINE config 1)?  usage: tools/jobs/r3_small.sh OUTDIR
out=$1; mkdir -p "$out"
b() { label=$1; shift
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs "$@" 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$label: %.0f Mpx/s  march %.4f ms  draw %.4f ms' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['draw_ms']))" | tee -a "$out/small.txt"; }
for round in 1 2; do
for sc in g8 g8x; do
  b "$sc 1080p/128 serial, default (no culling below 12 leaves)" --scene $sc --max-iter 128 --frames-in-flight 1
  b "$sc 1080p/128 serial, culling on (--prune 1)" --scene $sc --max-iter 128 --frames-in-flight 1 --prune 1
  b "$sc 1080p/128 four frames in flight, default" --scene $sc --max-iter 128
  b "$sc 1080p/128 four frames in flight, culling on" --scene $sc --max-iter 128 --prune 1
done
done
