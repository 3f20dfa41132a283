#!/bin/bash
# GPU-box job (round 3): kernels that do not cull -- refill lane by lane (default) or march in step, now that a batch is a 2x2 block of pixels?
out=$1; mkdir -p "$out"
b() { label=$1; shift
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs "$@" 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$label: %.0f Mpx/s  march %.4f ms  draw %.4f ms' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['draw_ms']))" | tee -a "$out/small.txt"; }
for round in 1 2; do
for sc in g8 g8x; do
  for rf in 0 16 32 64; do
    b "$sc 1080p/128 serial, refill threshold $rf" --scene $sc --max-iter 128 --frames-in-flight 1 --refill-min $rf
  done
done
for rf in 0 16 32 64; do
  b "xform_mix 1080p/256 serial, refill threshold $rf" --scene xform_mix --frames-in-flight 1 --refill-min $rf
done
done
