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
    """One call per leaf in program order, one combine per operator, operands resolved like the value stack
    of ray_marching.wgsl:187-203 would."""
    cc, w = serialize(oracle, scenes.g8())     # ((S u B) - S) u B
    body = [l.strip() for l in map_scene_body(renderer.jit_source(cc, w))]
    assert body == [
        "const float v0 = spec_sphere<FAST>(lp + 1, qx, qy, qz, tiny);",
        "const float v1 = spec_box<FAST>(lp + 9, qx, qy, qz, tiny);",
        "const float v2 = vmin(v0, v1);",
        "const float v3 = spec_sphere<FAST>(lp + 17, qx, qy, qz, tiny);",
        "const float v4 = vmax_negb(v2, v3);",
        "const float v5 = spec_box<FAST>(lp + 25, qx, qy, qz, tiny);",
        "const float v6 = vmin(v4, v5);",
        "return v6;",
    ]


def evaluate_generated(body, nodes_params, pos):
    """Interpret the generated straight-line code with plain float64 SDFs: a structural check only."""
    import math
    env = {}
    for line in body:
        line = line.strip().rstrip(";")
        if line.startswith("return"):
            return env[line.split()[1]]
        name, expr = line[len("const float "):].split(" = ")
        m = re.match(r"(\w+)(?:<FAST>)?\((.*)\)", expr)
        fn, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
        if fn in ("spec_sphere", "spec_box"):
            rec = (int(args[0].split("+")[1]) - 1) // 8
            p = nodes_params[rec]
            d = [pos[i] - p[i] for i in range(3)]
            if fn == "spec_sphere":
                env[name] = math.sqrt(sum(x * x for x in d)) - p[3]
            else:
                q = [abs(d[i]) - p[3 + i] for i in range(3)]
                env[name] = math.sqrt(sum(max(x, 0.0) ** 2 for x in q)) + min(max(q), 0.0)
        elif fn == "vmin":
            env[name] = min(env[args[0]], env[args[1]])
        elif fn == "vmax_negb":
            env[name] = max(env[args[0]], -env[args[1]])
        else:
            raise AssertionError(fn)
    raise AssertionError("no return")


@pytest.mark.parametrize("name", ["g8", "g32", "g32_balanced", "right_deep6"])
def test_generated_code_evaluates_like_the_oracle(oracle, name):
    scene = scenes.right_deep(6) if name == "right_deep6" else scenes.SCENES[name]()
    cc, w = serialize(oracle, scene)
    body = map_scene_body(renderer.jit_source(cc, w))
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
        got = evaluate_generated(body, params, [float(x) for x in pos])
        want = oracle.map_scene(cc, w, [float(np.float32(x)) for x in pos])
        assert abs(got - want) < 1e-4 * max(1.0, abs(want))


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


@pytest.mark.parametrize("name", ["g8", "g64", "ext_mix"])
def test_generated_kernel_compiles_for_gfx950(oracle, name):
    scene = (scenes.SCENES.get(name) or scenes.EXT_SCENES[name])()
    cc, w = serialize(oracle, scene)
    rc, ms, nbytes, log = renderer.jit_compile(cc, w)
    if rc != _ffi.RM_OK and "could not be loaded" in log:
        pytest.skip("libhiprtc is not installed: " + log)
    assert rc == _ffi.RM_OK, log
    assert nbytes > 4096 and ms > 0
