#!/bin/bash
# GPU-box job: vector instructions per wave-iteration of the march kernel for scenes of 1 / 4 / 16 primitives: what an iteration
# costs besides its leaves.  usage: tools/jobs/overhead.sh OUTDIR
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sc in g1 g8 g32; do
  echo "== $sc" >> "$out/overhead.txt"
  python3 tools/wave_stats.py --scene $sc --width 1920 --height 1080 --max-iter 256 --balance 3 $( [ $sc = g32 ] && echo --prune ) 2>/dev/null | grep -E "iterations:|lane occupancy" >> "$out/overhead.txt"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$out/pmc_$sc" -- python3 bench.py --scene $sc --steps 10 --warmup 2 --frames-in-flight 1 --no-cpu-baseline --no-legs > "$out/pmc_$sc.log" 2>&1
  python3 tools/pmc_summary.py "$out/pmc_$sc" 2>/dev/null | grep -A5 "rm_render_v5_spec" | grep -E "SQ_INSTS|mean=.*us" >> "$out/overhead.txt"
  rm -rf "$out/pmc_$sc"
done
cat "$out/overhead.txt"
