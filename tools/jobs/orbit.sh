#!/bin/bash
out=$1; mkdir -p "$out"
for sub in 1 0; do
  for steps in 20 200; do
    for cam in still orbit; do
      r=$(RM_CULL_SUBTRACTED=$sub python3 bench.py --camera $cam --steps $steps --warmup 5 --no-cpu-baseline --no-legs 2>>"$out/err.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f Mpx/s, ms/step %.3f draw %.3f ms march %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['draw_ms'], d['roofline']['kernel_ms']))")
      echo "subtracted leaves out of the tables=$sub | $cam camera, $steps steps | $r" | tee -a "$out/orbit.txt"
    done
  done
done
