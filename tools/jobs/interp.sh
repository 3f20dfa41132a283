#!/bin/bash
# GPU-box job: interpreter-kernel parity + timing (specialisation off).  usage: tools/jobs/interp.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not spec and (golden or ragged or limits or deep_stack or two_values or row_bands or batch_equals or culling or extension or fuzz or transform or materials or smooth)" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
for cfg in "g32 1920 1080 256" "g8 1920 1080 128" "g64 3840 2160 512" "g32_balanced 1920 1080 256"; do
  set -- $cfg
  for sp in 0 2; do
    timeout -k 10 200 python bench.py --scene $1 --width $2 --height $3 --max-iter $4 --specialize $sp --steps 30 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --no-legs > "$out/b_$1_$sp.json" 2>> "$out/bench.err"
    python3 -c "
import json,sys
d=json.load(open('$out/b_$1_$sp.json')); print('$1 $2x$3/$4 specialize=$sp: %.0f Mpx/s  march %.3f ms  draw %.3f ms' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['draw_ms']))"
  done
done
