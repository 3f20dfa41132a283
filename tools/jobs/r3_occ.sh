#!/bin/bash
# GPU-box job (round 3): (a) kernels above 80 vector registers forced to 6 waves per SIMD (RM_JIT_WAVES_PER_EU=6: they spill instead);
# (b) refill threshold sweep with lane occupancy (how much is there to recover from idle lanes).  usage: tools/jobs/r3_occ.sh OUTDIR
out=$1; mkdir -p "$out"
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for cfg in "g32s 3840 2160 256" "g32_balanced 1920 1080 256" "mat_mix 1920 1080 256" "xform_mix 1920 1080 256"; do set -- $cfg
    t "generated" $1 $2 $3 $4
    RM_JIT_WAVES_PER_EU=6 t "generated, 6 waves per SIMD forced" $1 $2 $3 $4
  done
done
for rf in 16 32 48 56 64; do
  echo "== refill threshold $rf" | tee -a "$out/refill.txt"
  python3 tools/wave_stats.py --scene g32 --balance 3 --refill-min $rf --prune 2>>"$out/err.log" | grep -E "kernel span|iterations:|lane occupancy" | tee -a "$out/refill.txt"
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --frames-in-flight 1 --refill-min $rf 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('   bench serial: draw %.4f ms kernel %.4f ms' % (d['ms_per_step'], d.get('kernel_ms', float('nan'))))" | tee -a "$out/refill.txt"
done
