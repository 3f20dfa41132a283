#!/bin/bash
# GPU-box job (round 3): 7 workgroups per CU (no partial normals in LDS for kernels with the four-tap function) -- parity subset, then A/B with
# the generated kernels capped at 72 vector registers (RM_JIT_WAVES_PER_EU=7).  usage: tools/jobs/r3_w7.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py tests/test_gpu_materials.py tests/test_gpu_fuzz.py -x -q -m gpu > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { tail -60 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for cfg in "g32 1920 1080 256" "g8 1920 1080 128" "g32 3840 2160 256" "g64 3840 2160 512" "g64 7680 4320 512" "g32_balanced 1920 1080 256" "g32s 3840 2160 256"; do set -- $cfg
    t "generated" $1 $2 $3 $4
    RM_JIT_WAVES_PER_EU=7 t "generated, 7 waves per SIMD forced" $1 $2 $3 $4
  done
done
for f in 1 4; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --frames-in-flight $f 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('bench F=$f: %.0f Mpx/s' % d['value'])" | tee -a "$out/times.txt"
RM_JIT_WAVES_PER_EU=7 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --frames-in-flight $f 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('bench F=$f, 7 waves forced: %.0f Mpx/s' % d['value'])" | tee -a "$out/times.txt"
done
cat "$out/status.txt"
