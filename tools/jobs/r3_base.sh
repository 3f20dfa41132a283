#!/bin/bash
# GPU-box job (round 3): the GPU suite with the round's new tests, bench as the driver runs it, kernel times of every config.
# usage: tools/jobs/r3_base.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$out/gpu_suite.log" 2>&1; echo "suite rc=$?" > "$out/status.txt"
tail -3 "$out/gpu_suite.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench1 rc=$?" >> "$out/status.txt"
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
done
cat "$out/status.txt"
