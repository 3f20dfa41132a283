#!/bin/bash
# GPU-box job: the whole GPU suite, a long fuzz run, the bench lines the driver will produce (N = 1, and two ranks sharing
# the GPU), and the kernel times of the configurations DESIGN.md quotes.  usage: tools/jobs/final.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$out/gpu_suite.log" 2>&1; echo "suite rc=$?" > "$out/status.txt"
tail -2 "$out/gpu_suite.log"
grep -q "suite rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
RM_FUZZ_FIRST_SEED=1000 RM_FUZZ_SEEDS=${FUZZ:-200} timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > "$out/fuzz.log" 2>&1; echo "fuzz rc=$?" >> "$out/status.txt"
tail -1 "$out/fuzz.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench1 rc=$?" >> "$out/status.txt"
timeout -k 10 300 python bench.py --gpus 2 --all-ranks-on-device0 --dist-backend gloo --steps 20 --warmup 5 > "$out/bench_n2.json" 2> "$out/bench_n2.err"; echo "bench2 rc=$?" >> "$out/status.txt"
timeout -k 10 300 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "benchdef rc=$?" >> "$out/status.txt"
t() {  # label scene w h iters extra-args..
  label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"
}
for round in 1 2; do
  t "generated" g32 1920 1080 256
  t "generated" g8 1920 1080 128
  t "generated" g32 3840 2160 256
  t "generated" g32s 3840 2160 256
  t "generated" g64 3840 2160 512
  t "generated" g64 7680 4320 512
  t "generated" mat_mix 1920 1080 256
  t "interpreter" g32 1920 1080 256 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
  t "interpreter" g64 7680 4320 512 --specialize 0
done
cat "$out/status.txt"
