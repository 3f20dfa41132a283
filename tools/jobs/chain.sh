#!/bin/bash
# GPU-box job: the interpreter kernels' chain loops (RM_CHAIN_MODE 0..3): parity subset, then timings.  usage: tools/jobs/chain.sh OUTDIR
out=$1; mkdir -p "$out"
sel="not spec and (golden or ragged or limits or deep_stack or two_values or row_bands or batch_equals or culling or extension or fuzz or transform or materials or smooth or grouped_far)"
for mode in ${MODES:-2}; do
  RM_CHAIN_MODE=$mode timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$sel" > "$out/tests_mode$mode.log" 2>&1; echo "tests mode $mode rc=$?" >> "$out/status.txt"
  tail -2 "$out/tests_mode$mode.log"
done
grep -q "rc=[1-9]" "$out/status.txt" && { cat "$out/status.txt"; exit 1; }
for cfg in "g32 1920 1080 256" "g8 1920 1080 128" "g64 3840 2160 512"; do
  set -- $cfg
  for mode in 0 1 2; do
    for k in 13 12; do
      [ $k = 12 ] && [ $mode != 2 ] && continue
      RM_CHAIN_MODE=$mode timeout -k 10 200 python bench.py --scene $1 --width $2 --height $3 --max-iter $4 --specialize 0 --kernel $k --steps 30 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --no-legs > "$out/b.json" 2>> "$out/bench.err"
      python3 -c "
import json
d=json.load(open('$out/b.json')); print('$1 $2x$3/$4 interpreter kernel $k loop $mode: %.0f Mpx/s  march %.3f ms  draw %.3f ms' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['draw_ms']))" | tee -a "$out/chain.txt"
    done
  done
done
cat "$out/status.txt"
