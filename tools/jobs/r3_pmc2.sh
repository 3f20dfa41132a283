#!/bin/bash
# GPU-box job: instruction mix of the march kernel (three rocprofv3 --pmc passes per configuration) + interpreter timings.
# usage: tools/jobs/r3_pmc2.sh OUTDIR
out=$1; mkdir -p "$out"
t() { label=$1; sc=$2; w=$3; h=$4; it=$5; shift 5
  r=$(python3 tools/time_kernel.py --scene $sc --width $w --height $h --max-iter $it --steps 30 "$@" 2>>"$out/err.log" | head -1)
  echo "$label | $sc ${w}x${h}/$it | $r" | tee -a "$out/times.txt"; }
for round in 1 2; do
  t "generated" g32 1920 1080 256
  t "interpreter" g32 1920 1080 256 --specialize 0
  t "interpreter" g32_balanced 1920 1080 256 --specialize 0
  t "interpreter" g32s 1920 1080 256 --specialize 0
  t "interpreter" g64 3840 2160 512 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32 1920 1080 256 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32_balanced 1920 1080 256 --specialize 0
  RM_CHAIN_MODE=1 t "interpreter, no unit masks" g32s 1920 1080 256 --specialize 0
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # label, bench args...
  label=$1; shift
  B="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs --frames-in-flight 1 $*"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$out/$label/pmc1" -- python3 $B > "$out/$label.pmc1.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/$label/pmc2" -- python3 $B > "$out/$label.pmc2.log" 2>&1
  python3 tools/pmc_summary.py "$out/$label" > "$out/$label.summary.txt"
  echo "== $label"; grep -A12 "rm_render_v5" "$out/$label.summary.txt" | grep -E "rm_render|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_BRANCH|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_LDS|GRBM"
}
prof g32 --scene g32
prof g32s --scene g32s
