#!/bin/bash
# GPU-box job: parity subset, then every BASELINE config on one GPU (serial loop, specialised kernel).
# usage: tools/jobs/configs.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "jit or fuzz or materials or golden or culling or extension or library_defaults or metric_config or smooth or transform or 4k_configs or baseline_config" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
for cfg in "g32 1920 1080 256" "g8 1920 1080 128" "g8x 1920 1080 128" "g32 3840 2160 256" "g32s 3840 2160 256" "g64 3840 2160 512" "g64 7680 4320 512" "mat_mix 1920 1080 256" "xform_mix 1920 1080 256"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --scene $1 --width $2 --height $3 --max-iter $4 --steps 30 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --no-legs > "$out/b_$1_$2.json" 2>> "$out/bench.err"
  python3 -c "
import json,sys
d=json.load(open('$out/b_$1_$2.json')); print('$1 $2x$3/$4: %.0f Mpx/s  march %.3f ms  draw %.3f ms' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['draw_ms']))" | tee -a "$out/configs.txt"
done
