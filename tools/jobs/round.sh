#!/bin/bash
# GPU-box job: parity subset, then march-kernel timings (tools/time_kernel.py) of the scenes that matter, generated
# and interpreter kernels.  usage: tools/jobs/round.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "jit or fuzz or materials or golden or culling or extension or library_defaults or metric_config or smooth or transform or deep_stack or ragged or limits or two_values" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
t() {  # label scene w h iters extra-args.. (env via leading VAR=..)
  label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"
}
for round in 1 2; do
  t "generated" g32 1920 1080 256
  t "generated" g64 3840 2160 512
  t "generated" g64 7680 4320 512
  t "generated" g32s 3840 2160 256
  t "generated" g8 1920 1080 128
  t "generated" mat_mix 1920 1080 256
  RM_JIT_MATERIAL_WALK=0 t "generated, interpreted material walk" mat_mix 1920 1080 256
  t "generated, tags stripped" mat_mix 1920 1080 256 --strip-tags
  t "interpreter" g32 1920 1080 256 --specialize 0
  t "interpreter" g8 1920 1080 128 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
  t "interpreter" g64 3840 2160 512 --specialize 0
done
