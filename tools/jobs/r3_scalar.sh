#!/bin/bash
# GPU-box job (round 3): fewer scalar instructions per march step (unit tests on a 32-bit word, groups of units behind one test,
# no diagnostics counters in the default kernel) -- parity tests first, then A/B timings.  usage: tools/jobs/r3_scalar.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_jit.py tests/test_gpu_fuzz_1080p.py -x -q -m gpu > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { tail -60 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for cfg in "g32 1920 1080 256" "g32_balanced 1920 1080 256" "g8 1920 1080 128" "g32 3840 2160 256" "g32s 3840 2160 256" "g64 3840 2160 512" "g64 7680 4320 512"; do set -- $cfg
    t "generated" $1 $2 $3 $4
    RM_JIT_UNIT_GROUPS=0 t "generated, no groups of units" $1 $2 $3 $4
    RM_JIT_UNIT_GROUPS=0 RM_JIT_UNIT_TEST=0 t "generated, no groups, unit tests on the 64-bit mask" $1 $2 $3 $4
  done
done
cat "$out/status.txt"
