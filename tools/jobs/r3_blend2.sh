#!/bin/bash
# GPU-box job (round 3): skip sets carried along the ray (blend programs) -- parity first, then timings.
# usage: tools/jobs/r3_blend2.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "local_skipping or smooth or lower_bounds or extension or fuzz or cull or pruning_policy or baseline_config or 4k_configs or materials" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; tail -40 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for sc in "g32s 3840 2160" "g32s 1920 1080"; do
    set -- $sc
    RM_BLEND_PRUNE_LEAVES=1000 t "plain (no rule)" $1 $2 $3 256
    RM_JIT_CACHED=0 t "local rule at every step" $1 $2 $3 256
    t "skip sets along the ray (default)" $1 $2 $3 256
    RM_JIT_BLEND_LEAF_TESTS=1 t "skip sets, members tested in a refresh" $1 $2 $3 256
    RM_BLEND_IN_STEP=0 t "skip sets, lanes refill one by one" $1 $2 $3 256
  done
done
for mode in 1 5; do
RM_JIT_PRUNE_STATS=$mode python3 tools/wave_stats.py --scene g32s --width 3840 --height 2160 --kernel 0 --prune --balance 3 > "$out/stats_mode$mode.txt" 2>&1
head -7 "$out/stats_mode$mode.txt"
done
cat "$out/status.txt"
