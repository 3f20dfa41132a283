#!/bin/bash
# GPU-box job: a parity subset and the march-kernel times of the main configurations.  usage: tools/jobs/quick.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "golden or metric_config or library_defaults or culling or lower_bounds or subtracted or fuzz or differential or strips or extension or 4k_configs" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -2 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; tail -30 "$out/tests.log"; exit 1; }
for round in 1 2; do
  for scene in "g32 1920 1080 256" "g8 1920 1080 128" "g64 3840 2160 512" "g32s 3840 2160 256" "g32 3840 2160 256"; do
    set -- $scene
    r=$(python3 tools/time_kernel.py --scene $1 --width $2 --height $3 --max-iter $4 --steps 30 2>>"$out/err.log" | head -1)
    echo "$1 $2x$3 | $r" | tee -a "$out/times.txt"
  done
done
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), round(d['one_frame_in_flight']['value']), round(d['end_to_end']['value']), round(d['orbit_camera']['value']), round(d['ab_interpreter_kernel']['value']))" | tee -a "$out/times.txt"; done
cat "$out/status.txt"
