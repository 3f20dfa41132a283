#!/bin/bash
# GPU-box job (round 3): chains capped at 72 registers + 6 workgroups per CU per launch as DEFAULTS -- parity first, then timings and the sweep.
out=$1; mkdir -p "$out"
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_1080p.py tests/test_gpu_materials.py -x -q -m gpu > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { tail -60 "$out/tests.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for cfg in "g32 1920 1080 256" "g8 1920 1080 128" "g8x 1920 1080 128" "g32 3840 2160 256" "g64 3840 2160 512" "g64 7680 4320 512" "g32_balanced 1920 1080 256" "g32s 3840 2160 256"; do set -- $cfg
    t "generated" $1 $2 $3 $4
  done
  t "interpreter" g32 1920 1080 256 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
done
echo "== sweep" | tee -a "$out/times.txt"; timeout -k 10 150 python3 tools/frames_in_flight_sweep.py 2>>"$out/err.log" | tee -a "$out/times.txt"
for f in 1 4; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --frames-in-flight $f 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('bench F=$f: %.0f Mpx/s' % d['value'])" | tee -a "$out/times.txt"
done
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-legs 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('bench F=4, 200 steps: %.0f Mpx/s' % d['value'])" | tee -a "$out/times.txt"
cat "$out/status.txt"
