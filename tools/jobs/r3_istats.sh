#!/bin/bash
# GPU-box job: how many records the interpreter's masked loops execute per evaluation.  usage: tools/jobs/r3_istats.sh OUTDIR
out=$1; mkdir -p "$out"
for sc in g32 g32_balanced; do
  echo "== $sc" | tee -a "$out/istats.txt"
  RM_HIP_SO=$PWD/build/variants/librm_hip_istats.so python3 tools/wave_stats.py --scene $sc --specialize 0 --interp-stats --balance 3 2>>"$out/err.log" | grep -v 'waves resident\|late wave' | tee -a "$out/istats.txt"
done
