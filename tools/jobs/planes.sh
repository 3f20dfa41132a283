#!/bin/bash
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lower_bounds or smooth or extension or culling or fuzz or config2_and_3 or subtracted or materials or transform" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -2 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; tail -40 "$out/tests.log"; exit 1; }
RM_CULL_SEEDS=200 timeout -k 10 900 python -m pytest tests/test_gpu_cull_differential.py -x -q -m gpu > "$out/diff.log" 2>&1; echo "diff rc=$?" >> "$out/status.txt"
tail -2 "$out/diff.log"
for bw in 1 0; do
  for scene in "ext_mix 1920 1080 256" "g8x 1920 1080 128" "g32s 3840 2160 256"; do
    set -- $scene
    r=$(RM_BOUND_WALK=$bw python3 tools/time_kernel.py --scene $1 --width $2 --height $3 --max-iter $4 --steps 30 2>>"$out/err.log" | head -1)
    echo "bound walk $bw | $1 $2x$3 | $r" | tee -a "$out/ab.txt"
  done
done
cat "$out/status.txt"
