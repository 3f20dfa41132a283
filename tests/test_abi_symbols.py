"""The C-ABI libraries load and export every symbol include/*.h declares; host-only entry
points behave without a GPU.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import scenes
from ray_marching_amd import _ffi, renderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rmh?_[a-z0-9_]+)\s*\(", text)))


def test_rm_abi_header_symbols_exported():
    names = declared_functions("rm_abi.h")
    assert len(names) >= 18
    L = _ffi.hip_lib()
    for n in names:
        assert hasattr(L, n), "librm_hip.so does not export %s" % n


def test_rm_host_header_symbols_exported():
    names = declared_functions("rm_host.h")
    assert len(names) >= 20
    L = _ffi.host_lib()
    for n in names:
        assert hasattr(L, n), "librm_host.so does not export %s" % n


def test_struct_sizes_match_reference_blobs():
    assert C.sizeof(_ffi.Uniforms) == 144 and C.sizeof(_ffi.Limits) == 12


def test_version_and_status_strings():
    L = _ffi.hip_lib()
    assert L.rm_abi_version() == 2
    for s in range(0, -12, -1):
        assert L.rm_status_string(s) not in (None, b"", b"unknown status")
    assert L.rm_status_string(-99) == b"unknown status"


def test_create_without_gpu_fails_loudly():
    L = _ffi.hip_lib()
    if L.rm_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_ffi.RmError) as e:
        renderer.RayMarchingResources(0)
    assert e.value.status == _ffi.RM_ERR_NO_DEVICE
    assert L.rm_create(0, None) == _ffi.RM_ERR_NULL


def test_null_context_is_an_error_not_a_crash():
    L = _ffi.hip_lib()
    assert L.rm_sync(None) == _ffi.RM_ERR_NULL
    assert L.rm_validate(None) == _ffi.RM_ERR_NULL
    assert L.rm_draw(None, 4, 4, 0, 4, None, 0, None) == _ffi.RM_ERR_NULL
    L.rm_destroy(None)


PROGRAM_CASES = [
    ("truncated sphere", 1, [0, 0, 0]),
    ("operator on empty stack", 1, [100]),
    ("operator with one operand", 2, [0, 0, 0, 0, 0x3F800000, 101]),
    ("unknown opcode", 1, [7]),
    ("reserved plane opcode", 1, [2]),
    ("cmd_count beyond words", 3, [0, 0, 0, 0, 0x3F800000]),
]


@pytest.mark.parametrize("label,cc,words", PROGRAM_CASES)
def test_validation_rejects_like_oracle_strict(oracle, label, cc, words):
    rc, _ = renderer.validate_program(cc, words)
    orc, _ = oracle.validate(cc, words, strict=True)
    assert rc < 0 and rc == orc, label


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_validation_accepts_scenes_and_reports_depth(oracle, name):
    cc, w = oracle.serialize(*scenes.SCENES[name]())
    assert renderer.validate_program(cc, w) == oracle.validate(cc, w, strict=True)


def test_validation_stack_limit(oracle):
    cc, w = oracle.serialize(*scenes.right_deep(32))
    assert renderer.validate_program(cc, w) == (0, 32)
    cc, w = oracle.serialize(*scenes.right_deep(33))
    assert renderer.validate_program(cc, w)[0] == _ffi.RM_ERR_STACK_OVERFLOW
    # left-deep chain of 33 primitives peaks at depth 2
    t = scenes._Tab()
    prims = [t.sphere((i, 0, 0), 0.4) for i in range(33)]
    cc, w = oracle.serialize(t.nodes, scenes._fold_left(t, prims))
    assert renderer.validate_program(cc, w) == (0, 2)


def test_validation_fuzz_agrees_with_oracle(oracle):
    rng = np.random.default_rng(11)
    ops = [0, 1, 100, 101, 100, 101, 2, 55]
    for _ in range(400):
        words = []
        n = int(rng.integers(0, 12))
        for _ in range(n):
            op = ops[int(rng.integers(0, len(ops)))]
            words.append(op)
            if op == 0:
                words += [0x3F000000] * 4
            elif op == 1:
                words += [0x3F000000] * 6
        if rng.random() < 0.3 and words:
            words = words[:int(rng.integers(0, len(words)))]
        cc = n if rng.random() < 0.8 else int(rng.integers(0, 14))
        assert renderer.validate_program(cc, words) == oracle.validate(cc, words, strict=True), (cc, words)
