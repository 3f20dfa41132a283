#!/bin/bash
# GPU-box job: subtracted primitives out of the miss-test tables: parity, then timings
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "subtracted or culling or lower_bounds or golden or metric_config or library_defaults or fuzz or extension or strips or 4k_configs or 8k_config or materials" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -2 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
for round in 1 2; do
  for scene in "g32 1920 1080 256" "g8 1920 1080 128" "g64 3840 2160 512" "g32s 3840 2160 256" "g32 3840 2160 256"; do
    set -- $scene
    r=$(python3 tools/time_kernel.py --scene $1 --width $2 --height $3 --max-iter $4 --steps 30 2>>"$out/err.log" | head -1)
    echo "$1 $2x$3 | $r" | tee -a "$out/times.txt"
  done
done
python bench.py --steps 20 --warmup 5 > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench rc=$?" >> "$out/status.txt"
python3 -c "
import json; d=json.load(open('$out/bench_n1.json')); print('bench', round(d['value']), d['one_frame_in_flight'], round(d['end_to_end']['value']), round(d['orbit_camera']['value']), round(d['ab_interpreter_kernel']['value']))"
cat "$out/status.txt"
