"""Structure specialiser (csrc/rm_jit.h), the part that needs no GPU: source generation from a command stream
and -- hipRTC cross-compiles like hipcc does -- compilation of the generated kernel for gfx950."""
import re

import numpy as np
import pytest

import scenes
from ray_marching_amd import _ffi, renderer


def serialize(oracle, scene):
    cc, w = oracle.serialize(*scene)
    return cc, np.asarray(w, dtype=np.uint32)


def map_scene_body(src):
    m = re.search(r"float map_scene_spec\(.*?\) \{\n(.*?)\n\}", src, re.S)
    assert m, src[-2000:]
    return m.group(1).splitlines()


def test_generated_code_follows_the_postfix_program(oracle):
    """One leaf evaluation per primitive in program order, one combine per operator, operands resolved like the
    value stack of ray_marching.wgsl:187-203 would; in the pruned form every sphere / box leaf sits behind its bit of the
    wave's unit mask."""
    cc, w = serialize(oracle, scenes.g8())     # ((S u B) - S) u B
    plain = [l.strip() for l in map_scene_body(renderer.jit_source(cc, w))]
    assert plain == [
        "const float x0 = qx, y0 = qy, z0 = qz;",
        "const float v0 = spec_sphere<FAST>(lp + 0, x0, y0, z0, tiny);",
        "const float v1 = vmin(v0, spec_box<FAST>(lp + 8, x0, y0, z0, tiny));",
        "const float v2 = vmax_negb(v1, spec_sphere<FAST>(lp + 16, x0, y0, z0, tiny));",
        "const float v3 = vmin(v2, spec_box<FAST>(lp + 24, x0, y0, z0, tiny));",
        "__builtin_amdgcn_sched_barrier(0);",
        "return v3;",
    ]
    body = [l.strip() for l in map_scene_body(renderer.jit_source(cc, w, prune=True))]
    assert body == [
        # every bounded leaf is a UNIT behind one bit of `need`, the mask wave-level culling computed for the whole wave
        # (rm_kernel_v5.h); skipped, a pushed leaf is +inf and a fused one leaves the accumulator alone.  Up to four consecutive
        # units -- a pushed leaf and the units that take its value on -- sit behind one more test on a word of the mask -- is ANY
        # of them needed --, which the tests inside share: most groups are skipped whole (rm_jit.h "GROUPS of units")
        "const float inf = __uint_as_float(0x7F800000u);",
        "const float x0 = qx, y0 = qy, z0 = qz;",
        "float v0 = inf;",
        "{ const uint32_t wg = unit_word(need, 0u);",
        "if ((wg & 0xfu) != 0u) {",
        "float v1 = inf;",
        "if (unit_in_word(wg, 0u)) {",
        "v1 = spec_sphere<FAST>(lp + 0, x0, y0, z0, tiny);",      # (records are staged rotated by one dword: parameters at dwords 0..6 of their 8)
        "}",
        "float v2 = v1;",
        "if (unit_in_word(wg, 1u)) {",
        "v2 = vmin(v1, spec_box<FAST>(lp + 8, x0, y0, z0, tiny));",
        "}",
        "float v3 = v2;",
        "if (unit_in_word(wg, 2u)) {",
        # a SUBTRACTED leaf: evaluated only if some live lane is inside it, or inside the accumulated solid (max(acc, -v) = acc else)
        "const float a = spec_sphere_a(lp + 16, x0, y0, z0);",
        "if (spec_sub_sphere_near(live, lp + 16, a, v2)) v3 = vmax_negb(v2, spec_sphere_v<FAST>(lp + 16, a, tiny));",
        "}",
        "float v4 = v3;",
        "if (unit_in_word(wg, 3u)) {",
        "v4 = vmin(v3, spec_box<FAST>(lp + 24, x0, y0, z0, tiny));",
        "}",
        "__builtin_amdgcn_sched_barrier(0);",
        "v0 = v4;",
        "} }",
        "return v0;",
    ]
    ungrouped = [l.strip() for l in map_scene_body(renderer.jit_source(cc, w, prune=True, env={"RM_JIT_UNIT_GROUPS": "0"}))]
    assert ungrouped[2:10] == ["float v0 = inf;", "if (unit_needed(need, 0u)) {", "v0 = spec_sphere<FAST>(lp + 0, x0, y0, z0, tiny);", "}",
                               "float v1 = v0;", "if (unit_needed(need, 1u)) {", "v1 = vmin(v0, spec_box<FAST>(lp + 8, x0, y0, z0, tiny));", "}"]
    assert ungrouped[-2:] == ["__builtin_amdgcn_sched_barrier(0);", "return v3;"]


def evaluate_generated(body, nodes_params, pos, prune_all_far=None):
    """Interpret the generated straight-line code with plain float64 SDFs: a structural check only.
    prune_all_far(value) -> bool decides whether a leaf counts as far (its unit's bit of `need` clear) for the (single) lane."""
    import math

    def sphere(rec):
        p = nodes_params[rec]
        return math.sqrt(sum((pos[i] - p[i]) ** 2 for i in range(3))) - p[3]

    def box(rec):
        p = nodes_params[rec]
        q = [abs(pos[i] - p[i]) - p[3 + i] for i in range(3)]
        return math.sqrt(sum(max(x, 0.0) ** 2 for x in q)) + min(max(q), 0.0)

    def leaf(kind, off):
        return sphere(int(off) // 8) if kind == "sphere" else box(int(off) // 8)

    ops = {"vmin": min, "vmax_negb": lambda a, b: max(a, -b), "fmax_": max}
    env = {"inf": math.inf}
    pending = None
    lines = [l.strip() for l in body]
    i = 0
    while i < len(lines):
        line = lines[i]
        i += 1
        if line.startswith(("const float inf", "const float x0", "__builtin_amdgcn_sched_barrier")) or line == "}":
            continue
        if line.startswith("return"):
            return env[line.rstrip(";").split()[1]]
        m = re.match(r"float (v\d+) = (\w+);", line)
        if m:
            env[m.group(1)] = env[m.group(2)]
            continue
        if line.startswith("{ const uint32_t wg = unit_word(need,"):      # a group of units: skipped whole when every unit in it is far
            assert re.match(r"if \(\(wg & 0x[0-9a-f]+u\) != 0u\) \{$", lines[i]), lines[i]
            end = lines.index("} }", i)
            inside = " ".join(lines[i + 1:end])
            vals = [leaf(k, off) for k, off in re.findall(r"spec_(sphere|box)(?:_a|<FAST>)\(lp \+ (\d+),", inside)]
            i = end + 1 if (prune_all_far and all(prune_all_far(v) for v in vals)) else i + 1
            continue
        if line == "} }":
            continue
        m = re.match(r"(v\d+) = (v\d+);$", line)
        if m:             # the value that leaves a group
            env[m.group(1)] = env[m.group(2)]
            continue
        m = re.match(r"if \((?:unit_needed\(need|unit_in_word\(wg), \d+u\)\) \{$", line)
        if m:             # one unit = one leaf: find its value, then decide whether the lane skips it
            block = []
            while lines[i] != "}":
                block.append(lines[i])
                i += 1
            text = " ".join(block)
            k = re.search(r"spec_(sphere|box)(?:_a|<FAST>)\(lp \+ (\d+),", text)
            val = leaf(k.group(1), k.group(2))
            if prune_all_far and prune_all_far(val):
                continue
            a = re.search(r"(v\d+) = (?:(\w+)\((v\d+), )?spec_(?:sphere|box)(?:_v)?<FAST>", text)
            tgt, op, acc = a.groups()
            env[tgt] = val if op is None else ops[op](env[acc], val)
            continue
        m = re.match(r"const float (v\d+) = (\w+)\((v\d+), (v\d+)\);", line)
        if m:
            env[m.group(1)] = ops[m.group(2)](env[m.group(3)], env[m.group(4)])
            continue
        raise AssertionError("unparsed line: " + line)
    raise AssertionError("no return")


@pytest.mark.parametrize("name", ["g8", "g32", "g32_balanced", "right_deep6"])
def test_generated_code_evaluates_like_the_oracle(oracle, name):
    scene = scenes.right_deep(6) if name == "right_deep6" else scenes.SCENES[name]()
    cc, w = serialize(oracle, scene)
    body = map_scene_body(renderer.jit_source(cc, w, prune=True))
    # parameters per decoded record: leaves in program order; fused operators share their leaf's record,
    # operators on sub-trees have a record of their own (rm_decode.h)
    params, ptr, i = [], 0, 0
    depth = 0
    while i < cc:
        op = int(w[ptr]); ptr += 1
        if op in (0, 1):
            n = 4 if op == 0 else 6
            params.append([float(x) for x in w[ptr:ptr + n].view(np.float32)])
            ptr += n
            if i + 1 < cc and depth >= 1 and int(w[ptr]) in (100, 101):
                ptr += 1
                i += 1
            else:
                depth += 1
        else:
            params.append(None)
            depth -= 1
        i += 1
    rng = np.random.default_rng(5)
    for pos in rng.uniform(-3, 3, size=(50, 3)):
        p = [float(x) for x in pos]
        got = evaluate_generated(body, params, p)
        want = oracle.map_scene(cc, w, [float(np.float32(x)) for x in pos])
        assert abs(got - want) < 1e-4 * max(1.0, abs(want))
        # the pruning rule: with |F| <= thr known, dropping every leaf whose value exceeds thr changes nothing
        for slack in (1.0, 1.5, 4.0):
            thr = abs(got) * slack + 1e-9
            assert evaluate_generated(body, params, p, prune_all_far=lambda v: v > thr) == got


def test_same_structure_same_source_different_structure_different_source(oracle):
    nodes, root = scenes.g8()
    cc, w = serialize(oracle, (nodes, root))
    moved = [(k, [x + 0.25 for x in p] if k in (0, 1) else p, l, r) for (k, p, l, r) in nodes]
    cc2, w2 = serialize(oracle, (moved, root))
    assert not np.array_equal(w, w2)
    assert renderer.jit_source(cc, w) == renderer.jit_source(cc2, w2)      # parameters are not baked in
    cc3, w3 = serialize(oracle, scenes.g32())
    assert renderer.jit_source(cc3, w3) != renderer.jit_source(cc, w)
    assert renderer.jit_source(cc, w, 2) != renderer.jit_source(cc, w, 4)   # waves per tile is part of the kernel


def test_source_rejects_what_cannot_be_specialised(oracle):
    with pytest.raises(_ffi.RmError) as e:
        renderer.jit_source(0, np.zeros(0, np.uint32))                       # empty program
    assert e.value.status == _ffi.RM_ERR_ARG
    with pytest.raises(_ffi.RmError) as e:
        renderer.jit_source(1, np.array([100], np.uint32))                   # invalid program: the decoder's status
    assert e.value.status == _ffi.RM_ERR_STACK_UNDERFLOW
    cc, w = serialize(oracle, scenes.g8())
    with pytest.raises(_ffi.RmError):
        renderer.jit_source(cc, w, 3)


@pytest.mark.parametrize("prune", [False, True], ids=["plain", "prune"])
@pytest.mark.parametrize("name", ["g8", "g64", "ext_mix"])
def test_generated_kernel_compiles_for_gfx950(oracle, name, prune):
    scene = (scenes.SCENES.get(name) or scenes.EXT_SCENES[name])()
    cc, w = serialize(oracle, scene)
    rc, ms, nbytes, log = renderer.jit_compile(cc, w, prune=prune)
    if rc != _ffi.RM_OK and "could not be loaded" in log:
        pytest.skip("libhiprtc is not installed: " + log)
    assert rc == _ffi.RM_OK, log
    assert nbytes > 4096 and ms > 0


def test_tree_kernels_are_capped_at_80_registers_when_that_is_cheap(oracle, monkeypatch):
    """The kernel of a lattice TREE may need a few registers more than 6 waves per SIMD allow (the balanced tree of the metric scene's
    leaves: 84).  The compiler is asked for the capped form first and it is kept iff the code object's kernel descriptor shows at most
    32 bytes of scratch per lane (rm_jit.h compile_best, code_object_scratch_bytes); a chain is compiled for 7 waves, without a probe."""
    monkeypatch.setenv("RM_JIT_CACHE_DIR", "off")
    cc, w = serialize(oracle, scenes.g32_balanced())
    rc, ms, nbytes, log = renderer.jit_compile(cc, w, prune=True)
    if rc != _ffi.RM_OK and "could not be loaded" in log:
        pytest.skip("libhiprtc is not installed: " + log)
    assert rc == _ffi.RM_OK, log
    m = re.search(r"kernel capped at 80 vector registers \(6 waves per SIMD\): (\d+) bytes of scratch per lane", log)
    assert m and int(m.group(1)) <= 32, log
    monkeypatch.setenv("RM_JIT_PROBE_CAP", "0")
    rc, ms, nbytes, log = renderer.jit_compile(cc, w, prune=True)
    assert rc == _ffi.RM_OK and "capped" not in log
    monkeypatch.delenv("RM_JIT_PROBE_CAP")
    cc, w = serialize(oracle, scenes.g32())
    rc, ms, nbytes, log = renderer.jit_compile(cc, w, prune=True)
    assert rc == _ffi.RM_OK and "capped" not in log
    assert "amdgpu_waves_per_eu(7, 7)" in renderer.jit_source(cc, w, prune=True)


def test_disk_cache_of_compiled_kernels(oracle, tmp_path):
    """RM_JIT_CACHE_DIR: the second compilation of the same kernel by a NEW process is a file read; another structure
    or a corrupted file is compiled afresh.  (A subprocess per step: the library reads the variable at compile time
    and keeps no in-memory cache on this path, but a fresh process is what the cache is for.)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from ray_marching_amd import csg, renderer\n"
            "cc, w = csg.serialize(csg.scene(sys.argv[1]))\n"
            "rc, ms, n, log = renderer.jit_compile(cc, w)\n"
            "print(rc, n, 'loaded' if 'loaded from' in log else ('nohiprtc' if 'could not be loaded' in log else 'compiled'))\n"
            % (str(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))),
               str(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
    env = dict(__import__('os').environ, RM_JIT_CACHE_DIR=str(tmp_path))

    def run(scene):
        r = subprocess.run([sys.executable, "-c", code, scene], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        return r.stdout.split()

    first = run("g8")
    if first[2] == "nohiprtc":
        pytest.skip("libhiprtc is not installed")
    assert first[0] == "0" and first[2] == "compiled"
    files = sorted(p.name for p in tmp_path.iterdir())
    assert len(files) == 1 and files[0].endswith(".co")
    second = run("g8")
    assert second == ["0", first[1], "loaded"]
    assert run("g32")[2] == "compiled" and len(list(tmp_path.iterdir())) == 2
    good = (tmp_path / files[0]).read_bytes()
    assert good[:8] == b"RMJITCO\x01" and good[32:36] == b"\x7fELF"      # 32-byte header (length, checksum), then the code object
    assert int.from_bytes(good[8:16], "little") == len(good) - 32
    # a damaged entry -- garbage, a truncated file, one flipped payload bit (only the checksum can see that one) -- is
    # recognised when it is READ, dropped and replaced; a later process loads the rewritten file
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x10
    for damaged in (b"not a code object", good[:len(good) // 2], bytes(flipped), good + b"tail"):
        (tmp_path / files[0]).write_bytes(damaged)
        assert run("g8")[2] == "compiled"
        assert (tmp_path / files[0]).read_bytes() == good
        assert run("g8")[2] == "loaded"


def test_disk_cache_is_on_by_default_next_to_the_library(oracle):
    """Without RM_JIT_CACHE_DIR the compiled kernels live in jit_cache/ next to librm_hip.so (build() leaves the BASELINE scenes'
    kernels there); RM_JIT_CACHE_DIR=off compiles every time."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from ray_marching_amd import csg, renderer\n"
            "cc, w = csg.serialize(csg.scene('g8'))\n"
            "rc, ms, n, log = renderer.jit_compile(cc, w)\n"
            "print(rc, 'nohiprtc' if 'could not be loaded' in log else log.strip().splitlines()[-1] if log.strip() else 'compiled')\n" % root)

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k != "RM_JIT_CACHE_DIR"}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0, r.stderr
        return r.stdout.strip()

    first = run()
    if "nohiprtc" in first:
        pytest.skip("libhiprtc is not installed")
    second = run()
    assert second.startswith("0 loaded from ") and os.path.join("ray-marching_amd", "jit_cache", "rm_") in second, second
    assert run(RM_JIT_CACHE_DIR="off") == "0 compiled"
    assert run(RM_JIT_CACHE_DIR="") == "0 compiled"


def test_four_taps_in_one_pass_function(oracle):
    """Generated kernels take the four normal taps of a hit (wgsl:135-144) in one pass: every record applied to the four
    positions c + k_t eps, in the pruned form behind ONE far test at the hit position.  Programs with a SmoothUnion
    keep the one-position taps (four copies of its division cost too many registers)."""
    def taps_body(src):
        m = re.search(r"void map_scene_taps\(.*?\) \{\n(.*?)\n\}\n\}", src, re.S)
        assert m, src[-1500:]
        return [l.strip() for l in m.group(1).splitlines()]

    cc, w = serialize(oracle, scenes.g8())     # ((S u B) - S) u B
    src = renderer.jit_source(cc, w)
    assert "#define RM_JIT_TAPS4 1" in src
    body = taps_body(src)
    assert body[:4] == ["const float e = 0.0001f;",
                        "const float x0_0 = cx + e, x0_1 = cx - e, x0_2 = cx - e, x0_3 = cx + e;",
                        "const float y0_0 = cy - e, y0_1 = cy - e, y0_2 = cy + e, y0_3 = cy + e;",
                        "const float z0_0 = cz - e, z0_1 = cz + e, z0_2 = cz - e, z0_3 = cz + e;"]
    assert body[4:8] == ["const float v0_%d = spec_sphere<FAST>(lp + 0, x0_%d, y0_%d, z0_%d, tiny);" % (t, t, t, t) for t in range(4)]
    assert body[8] == "guard_fence(tiny);"       # pins the sqrt guard after every leaf (register pressure: rm_kernel_v5.h)
    assert body[9:13] == ["const float v1_%d = vmin(v0_%d, spec_box<FAST>(lp + 8, x0_%d, y0_%d, z0_%d, tiny));" % (t, t, t, t, t) for t in range(4)]
    assert body[-4:] == ["f[%d] = v3_%d;" % (t, t) for t in range(4)]
    pruned = taps_body(renderer.jit_source(cc, w, prune=True))
    k = pruned.index("float v0_0 = inf;")
    assert pruned[k:k + 17] == ["float v0_0 = inf;", "float v0_1 = inf;", "float v0_2 = inf;", "float v0_3 = inf;",      # the value of the group of four units
                                "{ const uint32_t wg = unit_word(need, 0u);", "if ((wg & 0xfu) != 0u) {",
                                "float v1_0 = inf;", "float v1_1 = inf;", "float v1_2 = inf;", "float v1_3 = inf;",
                                "if (unit_in_word(wg, 0u)) {",       # one bit of the wave's unit mask for the four taps of a leaf
                                "v1_0 = spec_sphere<FAST>(lp + 0, x0_0, y0_0, z0_0, tiny);",
                                "v1_1 = spec_sphere<FAST>(lp + 0, x0_1, y0_1, z0_1, tiny);",
                                "v1_2 = spec_sphere<FAST>(lp + 0, x0_2, y0_2, z0_2, tiny);",
                                "v1_3 = spec_sphere<FAST>(lp + 0, x0_3, y0_3, z0_3, tiny);", "}", "guard_fence(tiny);"]
    # transforms: every scope gets four positions; smooth unions: the four blends behind ONE blend-zone test
    cc, w = serialize(oracle, scenes.xform_mix())
    src = renderer.jit_source(cc, w)
    assert "#define RM_JIT_TAPS4 1" in src and "x1_3" in "\n".join(taps_body(src))
    for scene in (scenes.g32s(), scenes.ext_mix()):
        cc, w = serialize(oracle, scene)
        src = renderer.jit_source(cc, w)
        assert "#define RM_JIT_TAPS4 1" in src and "spec_smooth_union4(lp + " in "\n".join(taps_body(src))
