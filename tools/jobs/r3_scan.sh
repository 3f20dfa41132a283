#!/bin/bash
# GPU-box job (round 3): the prefix minima of a blending chain's mask as v_min_f32_dpp -- selftests + parity first, then timings.
out=$1; mkdir -p "$out"
timeout -k 10 700 python -m pytest tests/test_gpu_arithmetic.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_1080p.py -x -q -m gpu > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { tail -60 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  t "generated" g32s 3840 2160 256
  t "generated" g32s 1920 1080 256
  t "generated" g32 1920 1080 256
  t "interpreter" g32s 1920 1080 256 --specialize 0
done
cat "$out/status.txt"
