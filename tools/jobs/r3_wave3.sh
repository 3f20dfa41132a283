#!/bin/bash
# GPU-box job (round 3): parity subset, then A/B timings of the wave-level culling variants.  usage: tools/jobs/r3_wave3.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "culling_rules or fuzz or jit or library_defaults or metric_config or golden or interpreter_record or extension or cross_lane or program_change or frames_in_flight or batch" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; tail -60 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for sc in "g32 1920 1080 256" "g64 3840 2160 512" "g32_balanced 1920 1080 256"; do
    set -- $sc
    t "two-step masks (default)" $1 $2 $3 $4
    RM_TWO_STEPS=0 t "a mask per step" $1 $2 $3 $4
    RM_JIT_WAVES_PER_EU=6 t "two-step masks, 80 VGPRs forced" $1 $2 $3 $4
    RM_JIT_WAVES_PER_EU=6 RM_TWO_STEPS=0 t "a mask per step, 80 VGPRs forced" $1 $2 $3 $4
  done
  t "generated" g8 1920 1080 128
  t "generated" g32s 3840 2160 256
  RM_JIT_WAVES_PER_EU=6 t "generated, 80 VGPRs forced" g32s 3840 2160 256
  t "interpreter" g32 1920 1080 256 --specialize 0
  RM_TWO_STEPS=0 t "interpreter, a mask per step" g32 1920 1080 256 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32_balanced 1920 1080 256 --specialize 0
  t "interpreter" g32s 1920 1080 256 --specialize 0
done
for v in 1 0; do
RM_TWO_STEPS=$v RM_JIT_PRUNE_STATS=1 python3 tools/wave_stats.py --scene g32 --width 1920 --height 1080 --kernel 0 --prune --balance 3 > "$out/stats_g32_two$v.txt" 2>&1
grep "leaves evaluated\|lane occupancy" "$out/stats_g32_two$v.txt"
done
cat "$out/status.txt"
