#!/bin/bash
# GPU-box job (round 3): the evidence of the round's end state.  usage: tools/jobs/r3_final.sh OUTDIR
#   1. the whole GPU suite  2. march kernel of every configuration (tools/time_kernel.py)  3. the BASELINE configs through bench.py's serial loop
#   4. smoke()  5. the default bench line  6. rocprofv3 kernel trace + PMC passes of the serial loop (tools/profile_r03.sh)
out=$1; mkdir -p "$out"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > "$out/gpu_suite.log" 2>&1; echo "suite rc=$?" > "$out/status.txt"
tail -3 "$out/gpu_suite.log"
grep -q "suite rc=0" "$out/status.txt" || { cat "$out/status.txt"; tail -60 "$out/gpu_suite.log"; exit 1; }
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  for cfg in "g32 1920 1080 256" "g32_balanced 1920 1080 256" "g8 1920 1080 128" "g32 3840 2160 256" "g32s 1920 1080 256" "g32s 3840 2160 256" "g64 3840 2160 512" "g64 7680 4320 512" "mat_mix 1920 1080 256" "xform_mix 1920 1080 256"; do set -- $cfg
    t "generated" $1 $2 $3 $4
  done
  for cfg in "g32 1920 1080 256" "g32_balanced 1920 1080 256" "g8 1920 1080 128" "g32s 1920 1080 256" "g64 3840 2160 512"; do set -- $cfg
    t "interpreter" $1 $2 $3 $4 --specialize 0
  done
  RM_JIT_WAVES_PER_EU=6 t "generated, 6 waves per SIMD forced" g32_balanced 1920 1080 256
done
for cfg in "g32 1920 1080 256" "g8 1920 1080 128" "g8x 1920 1080 128" "g32 3840 2160 256" "g32s 3840 2160 256" "g64 3840 2160 512" "g64 7680 4320 512" "xform_mix 1920 1080 256"; do set -- $cfg
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --frames-in-flight 1 --scene $1 --width $2 --height $3 --max-iter $4 2>>"$out/err.log" | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1 $2x$3/$4: %.0f Mpx/s  march %.3f ms  draw %.3f ms' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['draw_ms']))" | tee -a "$out/configs.txt"
done
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?" >> "$out/status.txt"; tail -2 "$out/smoke.log"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench rc=$?" >> "$out/status.txt"
python3 -c "
import json; d=json.load(open('$out/bench_n1.json'))
print('bench', d['value'], d['parity']['pixels_differing'], d.get('one_frame_in_flight'), d['ab_interpreter_kernel']['value'], d['roofline'])"
timeout -k 10 900 bash tools/profile_r03.sh "$out/prof" > "$out/prof.log" 2>&1; echo "prof rc=$?" >> "$out/status.txt"
grep -A12 "rm_render_v5_spec" "$out/prof/summary.txt" | head -60
cat "$out/status.txt"
