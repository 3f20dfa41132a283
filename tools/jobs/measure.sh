#!/bin/bash
# GPU-box job: parity subset, bench at N=1, one PMC pass of the serial loop.  usage: tools/jobs/measure.sh OUTDIR [pytest -k expr]
out=$1; kexpr=${2:-"jit or fuzz or materials or golden or culling or extension or library_defaults or metric_config"}
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$kexpr" > "$out/tests.log" 2>&1; echo "tests rc=$?" > "$out/status.txt"
tail -3 "$out/tests.log"
grep -q "tests rc=0" "$out/status.txt" || { cat "$out/status.txt"; exit 1; }
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench rc=$?" >> "$out/status.txt"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc" -- python3 bench.py --steps 20 --warmup 3 --frames-in-flight 1 --no-cpu-baseline --no-legs > "$out/pmc.log" 2>&1
python3 tools/pmc_summary.py "$out/pmc" > "$out/pmc_summary.txt"
cat "$out/status.txt"
python3 - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_n1.json"))
print("value %.0f  serial %.0f (kernel %.3f ms, draw %.3f ms)  orbit %.0f  interp %.0f  e2e %.0f" % (
    d["value"], d["one_frame_in_flight"]["value"], d["one_frame_in_flight"]["kernel_ms"], d["one_frame_in_flight"]["draw_ms"],
    d["orbit_camera"]["value"], d.get("ab_interpreter_kernel", {}).get("value", 0), d["end_to_end"]["value"]))
PY
grep -A9 "pmc.*rm_render_v5" "$out/pmc_summary.txt" | head -24
