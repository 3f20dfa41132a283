#!/bin/bash
# GPU-box job (round 3): the 72-register kernel with SIX workgroups per CU per launch (the seventh slot is then free for the next frame's launch).
out=$1; mkdir -p "$out"
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  t "default (73 registers, 7 workgroups per CU launched, 6 fit)" g32 1920 1080 256
  RM_WG_PER_CU_CAP=6 t "73 registers, 6 workgroups per CU launched" g32 1920 1080 256
  RM_JIT_WAVES_PER_EU=7 t "72 registers, 7 workgroups per CU" g32 1920 1080 256
  RM_JIT_WAVES_PER_EU=7 RM_WG_PER_CU_CAP=6 t "72 registers, 6 workgroups per CU launched" g32 1920 1080 256
  t "default" g32 3840 2160 256
  RM_JIT_WAVES_PER_EU=7 RM_WG_PER_CU_CAP=6 t "72 registers, 6 workgroups per CU launched" g32 3840 2160 256
done
echo "== sweep, default" | tee -a "$out/times.txt"; timeout -k 10 150 python3 tools/frames_in_flight_sweep.py 2>>"$out/err.log" | tee -a "$out/times.txt"
echo "== sweep, 72 registers, 6 workgroups per CU launched" | tee -a "$out/times.txt"; RM_JIT_WAVES_PER_EU=7 RM_WG_PER_CU_CAP=6 timeout -k 10 150 python3 tools/frames_in_flight_sweep.py 2>>"$out/err.log" | tee -a "$out/times.txt"
echo "== sweep, 73 registers, 6 workgroups per CU launched" | tee -a "$out/times.txt"; RM_WG_PER_CU_CAP=6 timeout -k 10 150 python3 tools/frames_in_flight_sweep.py 2>>"$out/err.log" | tee -a "$out/times.txt"
