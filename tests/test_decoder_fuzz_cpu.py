"""The upload-time decoder (ray-marching_amd/csrc/rm_decode.h) under AddressSanitizer + UndefinedBehaviorSanitizer: it is the one
place where bytes from the host application are interpreted.  tests/cpp/decode_fuzz.cpp feeds it well-formed random programs
(every node type, transform scopes, material tags, NaN / inf parameters), the same with words flipped, tails cut and command
counts off by a few, and pure noise; every stream must be rejected with a status or decode into records that satisfy the
invariants the kernels rely on (table slots within their tables, flags only where they mean something, depths within the
machine's limits).  CPU only: GPU sanitizers are not available on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decoder_survives_random_and_damaged_programs(tmp_path):
    cxx = os.environ.get("CXX", "g++")
    if not shutil.which(cxx):
        pytest.skip("no C++ compiler")
    exe = tmp_path / "decode_fuzz"
    build = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                            "-I", os.path.join(ROOT, "ray-marching_amd", "csrc"), "-o", str(exe),
                            os.path.join(ROOT, "tests", "cpp", "decode_fuzz.cpp")], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe), "30000"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "decoder fuzz ok" in run.stdout


def test_masked_tree_loop_equals_the_whole_program_with_the_unneeded_leaves_at_infinity(tmp_path):
    """tests/cpp/tree_keep_model.cpp: the interpreter's masked tree loop (rm_kernel_v5.h tree_keep, rm_interp.h
    map_scene_tree_masked) as a CPU model over the decoder's real per-record tables and the decision functions the kernel
    itself uses (rm_device.h rm_tree_*): for random trees, masks and leaf values, the records a mask leaves give the value of
    the whole program with the other leaves at +inf, bit for bit, on a stack one slot deeper than the program's own."""
    cxx = os.environ.get("CXX", "g++")
    if not shutil.which(cxx):
        pytest.skip("no C++ compiler")
    exe = tmp_path / "tree_keep_model"
    build = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                            "-I", os.path.join(ROOT, "ray-marching_amd", "csrc"), "-o", str(exe),
                            os.path.join(ROOT, "tests", "cpp", "tree_keep_model.cpp")], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe), "2000"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "tree keep model ok" in run.stdout


def test_elf_reader_of_the_specialiser_survives_damaged_code_objects(tmp_path):
    """tests/cpp/elf_scratch_fuzz.cpp: rm_jit.h code_object_scratch_bytes -- the reader that decides whether a register-capped kernel is
    kept, and that sees files from the on-disk kernel cache -- on a synthetic ELF64, every truncation of it, every byte damaged and
    20 000 noisy variants, under ASan + UBSan: a value or UINT32_MAX, never a read outside the buffer."""
    cxx = os.environ.get("CXX", "g++")
    if not shutil.which(cxx):
        pytest.skip("no C++ compiler")
    gen = os.path.join(ROOT, "ray-marching_amd", "csrc", "generated", "rm_jit_sources.inc")
    if not os.path.exists(gen):
        from ray_marching_amd import build
        build.generate_jit_sources()
    exe = tmp_path / "elf_scratch_fuzz"
    b = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ray-marching_amd", "csrc"), "-o", str(exe),
                        os.path.join(ROOT, "tests", "cpp", "elf_scratch_fuzz.cpp"), "-ldl", "-lpthread"], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "elf reader fuzz ok" in run.stdout
