#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the summary of a tools/profile_r02.sh run (tools/pmc_summary.py output).
usage: tools/update_pmc_json.py gpurun_out/<run>/summary.txt"""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary = open(sys.argv[1]).read()
path = os.path.join(root, "profiles", "pmc_traffic.json")
doc = json.load(open(path))
kernel, counters, durations, grbm = None, {}, {}, {}
for line in summary.splitlines():
    m = re.search(r"/(\w+)/[^/]+/\d+_counter_collection.csv\s+\[(?:rmk::)?(\w+)", line)
    if m:
        run, kernel = m.group(1), m.group(2)
        continue
    m = re.search(r"/(\w+)/[^/]+/\d+_kernel_trace.csv\s+\[(?:rmk::)?(\w+).*mean=([\d.]+) us", line)
    if m:
        durations[(m.group(1), m.group(2))] = float(m.group(3))
        continue
    m = re.match(r"\s+(\w+)\s+n=\d+\s+mean=([\d.e+]+)", line)
    if m and kernel:
        counters.setdefault(kernel, {})[m.group(1)] = float(m.group(2))
        if m.group(1) == "GRBM_GUI_ACTIVE":
            grbm.setdefault(kernel, {})[run] = float(m.group(2))
detail = {}
total = 0.0
for k in ("rm_tile_pre_v5", "rm_tile_sort_v5", "rm_render_v5_spec"):
    c = counters[k]
    detail[k] = {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"]}
    total += (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
doc["g32_1920x1080_256"] = int(round(total))
doc["g32_1920x1080_256_detail"] = detail
v = doc["g32_1920x1080_256_valu"]
c = counters["rm_render_v5_spec"]
for key in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
            "SQ_INSTS_VALU_INT32", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"):
    v[key] = c[key]
# the clock of the pass that counted GRBM_GUI_ACTIVE (8 XCDs), over that pass's own kernel duration (PMC passes serialise the
# kernels: their durations are a few % above the plain trace's)
run = sorted(grbm["rm_render_v5_spec"])[0]
v["GRBM_GUI_ACTIVE"] = grbm["rm_render_v5_spec"][run]
v["kernel_us_in_that_pass"] = durations[(run, "rm_render_v5_spec")]
v["clock_ghz"] = round(v["GRBM_GUI_ACTIVE"] / 8.0 / (v["kernel_us_in_that_pass"] * 1e3), 3)
v["lane_occupancy"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 3) if "SQ_THREAD_CYCLES_VALU" in c else v.get("lane_occupancy")
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps({"bytes": doc["g32_1920x1080_256"], "valu": v["SQ_INSTS_VALU"], "clock_ghz": v["clock_ghz"], "kernel_us": v["kernel_us_in_that_pass"],
                  "lane_occupancy": v["lane_occupancy"]}))
