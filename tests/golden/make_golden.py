"""Regenerates tests/golden/*: golden input/output vectors produced by the C oracle
(oracle/rm_oracle.c) in the build container.  The reference itself (Rust + WGSL) cannot be
executed there and ships no fixtures, so these vectors pin oracle <-> kernel and
oracle-version <-> oracle-version; they do NOT certify equality with a real wgpu render
("parity unpinned").

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenes  # noqa: E402
from oracle import cbind  # noqa: E402

CASES = [
    # name, scene, W, H, limits, camera events
    ("g1_64", "g1", 64, 64, (0.01, 100.0, 64), []),
    ("g8_64", "g8", 64, 64, (0.01, 100.0, 128), scenes.STILL_CAMERA_EVENTS),
    ("g32_64", "g32", 64, 64, (0.01, 100.0, 256), scenes.STILL_CAMERA_EVENTS),
    ("g32_balanced_64", "g32_balanced", 64, 64, (0.01, 100.0, 256), scenes.STILL_CAMERA_EVENTS),
    ("g64_48x40", "g64", 48, 40, (0.01, 100.0, 512), scenes.STILL_CAMERA_EVENTS),
    ("empty_40x24", None, 40, 24, (0.01, 100.0, 37), scenes.STILL_CAMERA_EVENTS),
    ("g8_inside_56", "g8", 56, 56, (0.01, 100.0, 100), [(2, -95.0, 0.0)]),   # camera dollied inside the solid
    # extension node types (semantics defined by this repo, not by the reference)
    ("g8x_64", "g8x", 64, 64, (0.01, 100.0, 128), scenes.STILL_CAMERA_EVENTS),
    ("g32s_64", "g32s", 64, 64, (0.01, 100.0, 256), scenes.STILL_CAMERA_EVENTS),
    ("ext_mix_64x48", "ext_mix", 64, 48, (0.01, 100.0, 128), scenes.STILL_CAMERA_EVENTS),
    ("xform_mix_64x48", "xform_mix", 64, 48, (0.01, 100.0, 128), scenes.STILL_CAMERA_EVENTS),   # space transformations
    ("mat_mix_64x48", "mat_mix", 64, 48, (0.01, 100.0, 128), scenes.STILL_CAMERA_EVENTS),       # material tags + table
]
# larger renders pinned by checksum + counters only
BIG = [
    ("g1_256", "g1", 256, 256, (0.01, 100.0, 64), []),
    ("g8_320x180", "g8", 320, 180, (0.01, 100.0, 128), scenes.STILL_CAMERA_EVENTS),
    ("g32_320x180", "g32", 320, 180, (0.01, 100.0, 256), scenes.STILL_CAMERA_EVENTS),
]


def inputs(scene, W, H, events):
    if scene is None:
        cc, words = 0, np.zeros(0, dtype=np.uint32)
    else:
        cc, words = cbind.serialize(*{**scenes.SCENES, **scenes.EXT_SCENES, **scenes.MAT_SCENES}[scene]())
    u, *_ = cbind.orbit_uniforms((float(W), float(H)), events=events)
    return cc, words, u


def main():
    index = {}
    for name, scene, W, H, lim, events in CASES + BIG:
        cc, words, u = inputs(scene, W, H, events)
        table = scenes.MATERIAL_TABLE if scene in scenes.MAT_SCENES else None
        img, cnt = cbind.render(u, lim, cc, words, W, H, threads=8, want_counters=True, materials=table)
        entry = {
            "scene": scene, "W": W, "H": H, "limits": list(lim), "cmd_count": cc,
            "words": [int(x) for x in words],
            "uniforms_u32": [int(x) for x in np.frombuffer(bytes(u), dtype=np.uint32)],
            "sha256": hashlib.sha256(img.tobytes()).hexdigest(),
            "counters": cnt,
        }
        if table is not None:
            entry["materials"] = [[float(np.float32(c)) for c in m] for m in table]
        if (name, scene, W, H, lim, events) in CASES:
            np.save(os.path.join(HERE, name + ".npy"), img)
            entry["file"] = name + ".npy"
        index[name] = entry
        print(name, entry["sha256"][:16], cnt)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
