#!/bin/bash
# GPU-box job (round 3): one lean interpreter kernel per record loop -- parity first, then timings.
out=$1; mkdir -p "$out"
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_jit.py tests/test_gpu_cull_differential.py tests/test_gpu_materials.py tests/test_gpu_fuzz_1080p.py -x -q -m gpu > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { tail -60 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  t "interpreter" g32 1920 1080 256 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
  t "interpreter" g8 1920 1080 128 --specialize 0
  t "interpreter" g64 3840 2160 512 --specialize 0
  t "interpreter" g32s 1920 1080 256 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32s 1920 1080 256 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32 1920 1080 256 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32_balanced 1920 1080 256 --specialize 0
  t "generated" g32 1920 1080 256
done
cat "$out/status.txt"
