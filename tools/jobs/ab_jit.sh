#!/bin/bash
# GPU-box job: A/B of generated-code knobs on ONE box, interleaved rounds (march kernel alone, tools/time_kernel.py).
# usage: tools/jobs/ab_jit.sh OUTDIR
out=$1; mkdir -p "$out"
run() {  # label, env...
  label=$1; shift
  for scene in "g32 1920 1080 256" "g64 3840 2160 512" "g32s 3840 2160 256"; do
    set -- $scene "$@"
    sc=$1; w=$2; h=$3; it=$4; shift 4
    r=$(env "$@" python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 2>>"$out/err.log" | head -1)
    echo "$label | $sc ${w}x${h} | $r" | tee -a "$out/ab.txt"
  done
}
for round in 1 2; do
  run "default                " RM_NOP=1
  run "group flags as bools   " RM_JIT_GROUP_MASK=0
  run "no guard fence         " RM_JIT_GUARD_FENCE=0
  run "fence in taps only     " RM_JIT_GUARD_FENCE=1
  run "inline slow taps (r1)  " RM_JIT_SLOW_TAPS_INLINE=1
  run "r1-like: all three     " RM_JIT_GROUP_MASK=0 RM_JIT_GUARD_FENCE=0 RM_JIT_SLOW_TAPS_INLINE=1
  run "smooth taps one by one " RM_JIT_TAPS4_SMOOTH=0
done
