#!/bin/bash
# GPU-box job (round 3): kernel trace + PMC passes (tools/profile_r03.sh) of config 3 (4K, blends) and of the interpreter kernel on the metric frame.
out=$1; mkdir -p "$out"
timeout -k 10 500 bash tools/profile_r03.sh "$out/g32s_4k" --scene g32s --width 3840 --height 2160 > "$out/g32s.log" 2>&1; echo "g32s rc=$?" | tee "$out/status.txt"
timeout -k 10 500 bash tools/profile_r03.sh "$out/interp" --specialize 0 > "$out/interp.log" 2>&1; echo "interp rc=$?" | tee -a "$out/status.txt"
grep -h -A10 "pmc[12]/.*rm_render_v5" "$out/g32s_4k/summary.txt" | grep -E "rm_render|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_BRANCH|SQ_INSTS_LDS|GRBM" | head -12
grep -h -A10 "pmc[12]/.*rm_render_v5" "$out/interp/summary.txt" | grep -E "rm_render|SQ_INSTS_VALU |SQ_INSTS_SALU|SQ_INSTS_BRANCH|SQ_INSTS_LDS|GRBM" | head -12
