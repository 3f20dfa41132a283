#!/bin/bash
# GPU-box job (round 3): wave-level culling -- the whole GPU suite, then timings of every configuration.
# usage: tools/jobs/r3_wave.sh OUTDIR
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
  t "generated" g32s 1920 1080 256
  t "generated" g64 3840 2160 512
  t "generated" g64 7680 4320 512
  t "interpreter" g32 1920 1080 256 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
  t "interpreter" g32s 1920 1080 256 --specialize 0
  RM_JIT_SUB_TESTS=0 t "generated, no subtractor tests" g32 1920 1080 256
  RM_JIT_SUB_TESTS=0 t "generated, no subtractor tests" g32s 3840 2160 256
done
for sc in g32 g32s; do
RM_JIT_PRUNE_STATS=1 python3 tools/wave_stats.py --scene $sc --width 1920 --height 1080 --kernel 0 --prune --balance 3 > "$out/stats_$sc.txt" 2>&1
head -6 "$out/stats_$sc.txt"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench1 rc=$?" >> "$out/status.txt"
cat "$out/status.txt"
