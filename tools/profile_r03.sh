#!/bin/bash
# Runs on the GPU box (via gpurun): round 3's kernel trace + PMC passes of the serial (one frame in flight) bench loop.
# Counter passes use --pmc with --kernel-trace only (no sys / hip traces); FETCH_SIZE and WRITE_SIZE in passes of their own.
# usage: tools/profile_r03.sh OUTDIR [bench.py args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --steps 40 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --no-legs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 $B > "$out/kt.log" 2>&1
B="bench.py --steps 20 --warmup 3 --frames-in-flight 1 --no-cpu-baseline --no-legs $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc1" -- python3 $B > "$out/pmc1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc2" -- python3 $B > "$out/pmc2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH \
  --output-format csv -d "$out/pmc3" -- python3 $B > "$out/pmc3.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $B > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 $B > "$out/pmc_write.log" 2>&1
python3 tools/pmc_summary.py "$out" > "$out/summary.txt"
cat "$out"/kt/*/*_kernel_stats.csv > "$out/kernel_stats.csv"
