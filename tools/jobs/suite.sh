#!/bin/bash
# GPU-box job: the whole GPU suite as the driver runs it, smoke(), then a few timings.  usage: tools/jobs/suite.sh OUTDIR
out=$1; mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$out/gpu_suite.log" 2>&1; echo "suite rc=$?" > "$out/status.txt"
tail -3 "$out/gpu_suite.log"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?" >> "$out/status.txt"; tail -1 "$out/smoke.log"
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
t "generated" mat_mix 1920 1080 256
RM_JIT_MATERIAL_WALK=0 t "generated, interpreted material walk" mat_mix 1920 1080 256
t "generated, tags stripped" mat_mix 1920 1080 256 --strip-tags
t "generated" xform_mix 1920 1080 256
t "generated" g32 1920 1080 256
cat "$out/status.txt"
