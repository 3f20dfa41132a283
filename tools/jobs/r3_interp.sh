#!/bin/bash
# GPU-box job (round 3): the whole GPU suite, then the interpreter kernels' loops, A/B.  usage: tools/jobs/r3_interp.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > "$out/gpu_suite.log" 2>&1; echo "suite rc=$?" > "$out/status.txt"
tail -3 "$out/gpu_suite.log"
grep -q "suite rc=0" "$out/status.txt" || { cat "$out/status.txt"; tail -60 "$out/gpu_suite.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  t "generated" g32 1920 1080 256
  t "generated" g32_balanced 1920 1080 256
  t "generated" g8 1920 1080 128
  t "generated" g32 3840 2160 256
  t "generated" g32s 3840 2160 256
  t "generated" g64 3840 2160 512
  t "generated" g64 7680 4320 512
  for sc in "g32 1920 1080 256" "g32_balanced 1920 1080 256" "g64 3840 2160 512" "g8 1920 1080 128"; do set -- $sc
    t "interpreter" $1 $2 $3 $4 --specialize 0
    RM_CHAIN_MODE=1 t "interpreter, no unit masks" $1 $2 $3 $4 --specialize 0
    RM_CHAIN_MODE=0 t "interpreter, general loop" $1 $2 $3 $4 --specialize 0
    RM_HIP_SO=$PWD/build/variants/librm_hip_lean5.so t "interpreter, 5 waves per SIMD allowed" $1 $2 $3 $4 --specialize 0
  done
  t "interpreter" g32s 1920 1080 256 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32s 1920 1080 256 --specialize 0
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench1 rc=$?" >> "$out/status.txt"
python3 -c "
import json; d=json.load(open('$out/bench_n1.json'))
print('bench', d['value'], d['parity']['pixels_differing'], d['one_frame_in_flight'], d['ab_interpreter_kernel']['value'])"
cat "$out/status.txt"
