#!/bin/bash
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$out/gpu_suite.log" 2>&1; echo "suite rc=$?" > "$out/status.txt"
tail -3 "$out/gpu_suite.log"
b() { label=$1; shift
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --no-legs "$@" > "$out/tmp.json" 2>> "$out/bench.err"
  python3 -c "
import json
d=json.load(open('$out/tmp.json')); print('$label: %.0f Mpx/s  march %.3f ms  draw %.3f ms' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['draw_ms']))" | tee -a "$out/interp.txt"; }
b "generated" 
b "interpreter lds wpt4" --specialize 0
b "interpreter smem wpt4" --specialize 0 --kernel 12
b "interpreter lds wpt8" --specialize 0 --waves-per-tile 8
b "interpreter lds wpt2" --specialize 0 --waves-per-tile 2
cat "$out/status.txt"
