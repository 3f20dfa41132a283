#!/bin/bash
# GPU-box job: generated march function with / without the per-leaf far tests inside near pairs (RM_JIT_LEAF_TESTS).
# usage: tools/jobs/leaf_tests.sh OUTDIR
out=$1; mkdir -p "$out"
for v in 0 2; do
  RM_JIT_LEAF_TESTS=$v timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "spec_prune or grouped_far or pruning_policy" > "$out/tests_$v.log" 2>&1; echo "tests leaf_tests=$v rc=$?" >> "$out/status.txt"
  tail -1 "$out/tests_$v.log"
done
grep -q "rc=[1-9]" "$out/status.txt" && { cat "$out/status.txt"; exit 1; }
for round in 1 2; do
  for v in 1 0 2 3; do
    for scene in "g32 1920 1080 256" "g64 3840 2160 512" "g32_balanced 1920 1080 256"; do
      set -- $scene
      r=$(RM_JIT_LEAF_TESTS=$v python3 tools/time_kernel.py --scene $1 --width $2 --height $3 --max-iter $4 --steps 30 2>>"$out/err.log" | head -1)
      echo "leaf tests $v | $1 $2x$3 | $r" | tee -a "$out/ab.txt"
    done
  done
done
cat "$out/status.txt"
