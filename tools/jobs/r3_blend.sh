#!/bin/bash
# GPU-box job (round 3): the local skipping rule of blending programs -- parity first, then timings with its knobs.
# usage: tools/jobs/r3_blend.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "local_skipping or smooth or lower_bounds or extension or fuzz or cull or pruning_policy or baseline_config or 4k_configs or materials" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for sc in "g32s 3840 2160" "g32s 1920 1080"; do
    set -- $sc
    RM_BLEND_PRUNE_LEAVES=1000 t "plain (no rule)" $1 $2 $3 256
    t "local rule, defaults" $1 $2 $3 256
    RM_BLEND_IN_STEP=0 t "local rule, lanes refill one by one" $1 $2 $3 256
    RM_JIT_BLEND_LEAF_TESTS=0 t "local rule, pairs only" $1 $2 $3 256
    RM_JIT_BLEND_LEAF_TESTS=2 t "local rule, pairs + boxes" $1 $2 $3 256
    RM_JIT_BLEND_LEAF_TESTS=3 t "local rule, pairs + spheres" $1 $2 $3 256
    RM_JIT_BLEND_UPFRONT=0 t "local rule, pair distances where used" $1 $2 $3 256
  done
  t "g8x default" g8x 1920 1080 128
  RM_BLEND_PRUNE_LEAVES=1 t "g8x local rule" g8x 1920 1080 128
  t "ext_mix default" ext_mix 1920 1080 256
  RM_BLEND_PRUNE_LEAVES=1 t "ext_mix local rule" ext_mix 1920 1080 256
done
RM_JIT_PRUNE_STATS=1 python3 tools/wave_stats.py --scene g32s --width 3840 --height 2160 --kernel 0 --prune --balance 3 > "$out/stats_leaves.txt" 2>&1
RM_JIT_PRUNE_STATS=1 RM_BLEND_IN_STEP=0 python3 tools/wave_stats.py --scene g32s --width 3840 --height 2160 --kernel 0 --prune --balance 3 > "$out/stats_leaves_nostep.txt" 2>&1
head -6 "$out/stats_leaves.txt" "$out/stats_leaves_nostep.txt"
cat "$out/status.txt"
