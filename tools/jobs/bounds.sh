#!/bin/bash
# GPU-box job: the miss test on lower bounds (programs with SmoothUnion): parity, then the smooth-min configuration with and without it
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lower_bounds or smooth or extension or culling or fuzz or 4k_configs or config2_and_3" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -2 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
for round in 1 2; do
  for bw in 1 0; do
    for scene in "g32s 3840 2160 256" "g32s 1920 1080 256" "g32 1920 1080 256"; do
      set -- $scene
      r=$(RM_BOUND_WALK=$bw python3 tools/time_kernel.py --scene $1 --width $2 --height $3 --max-iter $4 --steps 30 2>>"$out/err.log" | head -1)
      echo "bound walk $bw | $1 $2x$3 | $r" | tee -a "$out/ab.txt"
    done
    b=$(RM_BOUND_WALK=$bw python3 bench.py --scene g32s --width 3840 --height 2160 --steps 30 --warmup 5 --no-cpu-baseline --no-legs 2>>"$out/err.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f Mpx/s, draw %.3f ms march %.3f ms' % (d['value'], d['roofline']['draw_ms'], d['roofline']['kernel_ms']))")
    echo "bound walk $bw | bench g32s 4K four frames in flight | $b" | tee -a "$out/ab.txt"
  done
done
cat "$out/status.txt"
