#!/usr/bin/env python3
"""Registers, LDS and spills of every kernel, from the compiler (no GPU needed): the library's kernels through
`hipcc -Rpass-analysis=kernel-resource-usage`, the hipRTC-compiled march kernel of a scene through the metadata of its
code object (llvm-readelf --notes).  usage: tools/kernel_resources.py [scene ...]  > profiles/r02_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def library_kernels():
    from ray_marching_amd import build
    build.generate_jit_sources()
    cmd = [build.hipcc_path()] + [f for f in build.HIP_FLAGS if f not in ("-shared", "-fPIC")] + [
        "-I", build.INCLUDE, "-I", build.CSRC, "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
        "-o", os.devnull, os.path.join(build.CSRC, "rm_abi.hip")]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def jit_kernel(scene, prune):
    from ray_marching_amd import csg, renderer
    cc, words = csg.serialize(csg.scene(scene))
    with tempfile.TemporaryDirectory() as d:
        os.environ["RM_JIT_DUMP_DIR"] = d
        old_cache = os.environ.get("RM_JIT_CACHE_DIR")
        os.environ["RM_JIT_CACHE_DIR"] = "off"       # (a kernel read from the disk cache is not dumped)
        rc, ms, nbytes, log = renderer.jit_compile(cc, words, prune=prune)
        del os.environ["RM_JIT_DUMP_DIR"]
        if old_cache is None:
            del os.environ["RM_JIT_CACHE_DIR"]
        else:
            os.environ["RM_JIT_CACHE_DIR"] = old_cache
        if rc != 0:
            return None
        co = [f for f in os.listdir(d) if f.endswith(".co")][0]
        notes = subprocess.run([READELF, "--notes", os.path.join(d, co)], capture_output=True, text=True).stdout
    out = {}
    for key in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
                "private_segment_fixed_size", "kernarg_segment_size"):
        m = re.search(r"\.%s:\s+(\d+)" % key, notes)
        if m:
            out[key] = int(m.group(1))
    return out


def waves_per_simd(vgprs):
    alloc = (vgprs + 7) // 8 * 8            # MI355X_MICROARCH.md: allocation granule 8, 512 registers per lane per SIMD
    return min(8, 512 // max(alloc, 8)), alloc


def main():
    print("# Kernel resources from the compiler (ROCm %s); waves/SIMD = min(8, 512 / VGPRs rounded up to 8)" %
          (open("/opt/rocm/.info/version").read().strip() if os.path.exists("/opt/rocm/.info/version") else "?"))
    print("# rocprofv3's per-dispatch VGPR_Count column reads HALF these numbers on gfx950 (e.g. 40 for the 79-80 of the\n"
          "# specialised march kernel) and its LDS column shows the STATIC segment only: the march kernels take all their LDS\n"
          "# dynamically (hipModuleLaunchKernel sharedMemBytes; 23-26 KB per 4-wave workgroup for the metric scene).")
    print("\n## library kernels (hipcc, librm_hip.so)")
    for r in library_kernels():
        v = int(r.get("VGPRs", "0"))
        w, alloc = waves_per_simd(v)
        print("%-100s VGPRs %3d (alloc %3d -> %d waves/SIMD)  SGPRs %3s  spills s/v %s/%s  static LDS %s B  scratch %s B" % (
            demangle(r["name"])[:100], v, alloc, w, r.get("TotalSGPRs", "?"), r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"),
            r.get("LDS Size [bytes/block]", "?"), r.get("ScratchSize [bytes/lane]", "?")))
    print("\n## specialised march kernel rm_render_v5_spec (hipRTC, per scene structure)")
    for scene in (sys.argv[1:] or ["g8", "g32", "g64", "g32s", "xform_mix", "mat_mix"]):
        for prune in (False, True):
            m = jit_kernel(scene, prune)
            if not m:
                continue
            w, alloc = waves_per_simd(m.get("vgpr_count", 0))
            print("%-10s %-7s VGPRs %3d (alloc %3d -> %d waves/SIMD)  SGPRs %3d  spills s/v %d/%d  static LDS %d B  scratch %d B  kernarg %d B" % (
                scene, "pruned" if prune else "plain", m.get("vgpr_count", 0), alloc, w, m.get("sgpr_count", 0),
                m.get("sgpr_spill_count", 0), m.get("vgpr_spill_count", 0), m.get("group_segment_fixed_size", 0),
                m.get("private_segment_fixed_size", 0), m.get("kernarg_segment_size", 0)))


if __name__ == "__main__":
    main()
