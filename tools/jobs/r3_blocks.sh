#!/bin/bash
# GPU-box job (round 3): ray batches = the 16 samples of a 2x2 block of pixels -- parity tests first, then timings + lane occupancy.
# usage: tools/jobs/r3_blocks.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_jit.py tests/test_gpu_fuzz_1080p.py tests/test_gpu_materials.py tests/test_gpu_cull_differential.py -x -q -m gpu > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { tail -60 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for cfg in "g32 1920 1080 256" "g32_balanced 1920 1080 256" "g8 1920 1080 128" "g32 3840 2160 256" "g32s 3840 2160 256" "g64 3840 2160 512" "g64 7680 4320 512" "mat_mix 1920 1080 256" "xform_mix 1920 1080 256"; do set -- $cfg
    t "generated" $1 $2 $3 $4
  done
  t "interpreter" g32 1920 1080 256 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
  t "interpreter" g8 1920 1080 128 --specialize 0
done
python3 tools/wave_stats.py --scene g32 --balance 3 --prune 2>>"$out/err.log" | grep -E "kernel span|iterations:|lane occupancy" | tee -a "$out/occ.txt"
python3 tools/wave_stats.py --scene g64 --width 3840 --height 2160 --max-iter 512 --balance 3 --prune 2>>"$out/err.log" | grep -E "kernel span|iterations:|lane occupancy" | tee -a "$out/occ.txt"
cat "$out/status.txt"
