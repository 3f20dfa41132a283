#!/bin/bash
# GPU-box job: instruction mix of the march kernel, per configuration, from three rocprofv3 --pmc passes (no tracing besides).
# usage: tools/jobs/r3_pmc_blend.sh OUTDIR
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # label, bench args...
  label=$1; shift
  B="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs --frames-in-flight 1 $*"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$out/$label/pmc1" -- python3 $B > "$out/$label.pmc1.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/$label/pmc2" -- python3 $B > "$out/$label.pmc2.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH \
    --output-format csv -d "$out/$label/pmc3" -- python3 $B > "$out/$label.pmc3.log" 2>&1
  python3 tools/pmc_summary.py "$out/$label" > "$out/$label.summary.txt"
  echo "== $label"; grep -A12 "rm_render_v5" "$out/$label.summary.txt" | grep -E "rm_render|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_BRANCH|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|TRANS|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_VALU|GRBM" 
}
prof g32 --scene g32
export RM_BLEND_PRUNE_LEAVES=1000
prof g32s_plain --scene g32s
unset RM_BLEND_PRUNE_LEAVES
export RM_JIT_BLEND_LEAF_TESTS=0
prof g32s_pairs --scene g32s
unset RM_JIT_BLEND_LEAF_TESTS
prof g32s_all_tests --scene g32s
