"""bindings/rust (the reference's host language; reference host side: src/ray_marching/renderer.rs:184-256).

The build image has no rustc / cargo, so the crate cannot be compiled here.  What can be checked without a compiler
is the part that silently breaks a binding: every function of include/rm_abi.h must be declared in lib.rs's
`extern "C"` block with the same name, arity and parameter / return types; constants must carry the header's values;
attributes must sit on items that accept them (round 1 shipped a `#[derive(Debug)]` on a `const`).  When a Rust
toolchain IS on PATH the crate is also type-checked with `cargo check --offline`."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "rust")

C_TO_RUST = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "double": "f64",
             "float": "f32", "char": "c_char", "void": "c_void", "rm_ctx": "rm_ctx", "rm_uniforms": "rm_uniforms",
             "rm_limits": "rm_limits"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_type_to_rust(t):
    t = t.strip()
    const = False
    if t.startswith("const "):
        const, t = True, t[6:].strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    r = C_TO_RUST[base]
    for level in range(stars):
        # `const T*` is a pointer to const T; further levels (rm_ctx**) are pointers to mutable pointers
        r = ("*const " if (const and level == 0) else "*mut ") + r
    return r


def header_functions():
    text = strip_c_comments(open(os.path.join(ROOT, "include", "rm_abi.h")).read())
    text = re.sub(r"#.*", "", text)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(rm_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = []
        if params and params != "void":
            for p in params.split(","):
                p = " ".join(p.split())
                pm = re.match(r"(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", p)
                assert pm, p
                plist.append(c_type_to_rust(pm.group(1)))
        out[name] = (None if ret == "void" else c_type_to_rust(ret), plist)
    return out


def rust_source():
    return open(os.path.join(CRATE, "src", "lib.rs")).read()


def rust_functions():
    src = re.sub(r"//.*", "", rust_source())
    block = re.search(r'extern "C" \{(.*?)\n\}', src, re.S)
    assert block
    out = {}
    for m in re.finditer(r"pub fn (rm_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block.group(1), re.S):
        name, params, ret = m.group(1), " ".join(m.group(2).split()), m.group(3)
        plist = []
        if params:
            for p in params.split(","):
                p = p.strip()
                if p:
                    plist.append(p.split(":", 1)[1].strip())
        out[name] = (ret.strip() if ret else None, plist)
    return out


def test_extern_block_matches_the_header():
    c, r = header_functions(), rust_functions()
    assert len(c) >= 30
    assert sorted(c) == sorted(r), "missing in lib.rs: %s; not in rm_abi.h: %s" % (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in sorted(c):
        assert c[name] == r[name], "%s: header %s, lib.rs %s" % (name, c[name], r[name])


def header_constants():
    text = strip_c_comments(open(os.path.join(ROOT, "include", "rm_abi.h")).read())
    consts = {}
    for body in re.findall(r"enum\s+rm_[a-z]+\s*\{(.*?)\}", text, re.S):
        for name, value in re.findall(r"(RM_[A-Z0-9_]+)\s*=\s*(-?\d+)", body):
            consts[name] = int(value)
    consts["RM_ABI_VERSION"] = int(re.search(r"#define RM_ABI_VERSION (\d+)", text).group(1))
    consts["RM_JIT_PRUNE"] = int(re.search(r"#define RM_JIT_PRUNE (0x[0-9a-fA-F]+)", text).group(1), 16)
    return consts


def test_constants_carry_the_header_values():
    c = header_constants()
    r = {n: int(v, 0) for n, v in re.findall(r"pub const (RM_[A-Z0-9_]+): c_int = (-?(?:0x)?[0-9a-fA-F]+);", rust_source())}
    assert len(r) >= 45
    for name, value in r.items():
        assert name in c, "%s is not in rm_abi.h" % name
        assert c[name] == value, "%s: header %d, lib.rs %d" % (name, c[name], value)
    missing = sorted(set(c) - set(r))
    assert not missing, "constants of rm_abi.h without a Rust counterpart: %s" % missing


def test_attributes_sit_on_items_that_accept_them():
    lines = rust_source().splitlines()
    for i, line in enumerate(lines):
        s = line.strip()
        if not s.startswith("#[") or s.startswith("#[cfg(test)]") or s.startswith("#[test]"):
            continue
        j = i + 1
        while lines[j].strip().startswith(("#[", "///", "//")) or not lines[j].strip():
            j += 1
        item = lines[j].strip()
        if s.startswith("#[derive") or s.startswith("#[repr"):
            assert re.match(r"pub (struct|enum|union) ", item), "line %d: %s on `%s`" % (i + 1, s, item)
        elif s.startswith("#[link"):
            assert item.startswith('extern "C"'), "line %d: %s on `%s`" % (i + 1, s, item)
    src = rust_source()
    assert re.search(r"#\[derive\(Debug[^)]*\)\]\s*pub struct RmError", src)     # Result<_, RmError>::unwrap() needs it
    assert "impl fmt::Display for RmError" in src and "impl std::error::Error for RmError" in src
    assert src.count("{") == src.count("}") and src.count("(") == src.count(")")


def test_safe_wrappers_check_slice_lengths_before_the_library_writes():
    """A safe fn that hands `out_rgba.as_mut_ptr()` to the library must first assert that the slice holds what the library
    writes (an unsound API otherwise).  The crate cannot be compiled here, so this is a text check per wrapper -- and the
    Rust strip_row_count is re-evaluated in Python against the library's own count."""
    src = rust_source()
    for fn in re.finditer(r"pub fn (\w+)\(&self[^{]*out_rgba: &mut \[f32\][^{]*\{(.*?)\n    \}\n", src, re.S):
        body = fn.group(2)
        assert "out_rgba.as_mut_ptr()" in body
        assert re.search(r"assert!\(out_rgba\.len\(\) >= ", body[: body.index("out_rgba.as_mut_ptr()")]), fn.group(1)
    assert len(re.findall(r"pub fn \w+\(&self[^{]*out_rgba: &mut \[f32\]", src)) >= 2          # draw, draw_strips
    m = re.search(r"pub fn strip_row_count\(height: u32, strip_rows: u32, first: u32, stride: u32\) -> u32 \{(.*?)\n\}\n", src, re.S)
    assert m
    def rust_count(height, strip_rows, first, stride):     # the function above, statement for statement
        if strip_rows == 0 or stride == 0 or first >= stride:
            return 0
        n_strips = (height + strip_rows - 1) // strip_rows
        rows, s = 0, first
        while s < n_strips:
            rows += min(height - s * strip_rows, strip_rows)
            s += stride
        return rows
    for stmt in ("let n_strips = (height + strip_rows - 1) / strip_rows;", "rows += (height - r0).min(strip_rows);", "s += stride;"):
        assert stmt in m.group(1), stmt
    from ray_marching_amd import shard
    for H in (1, 15, 16, 17, 1080, 2160):
        for sr in (8, 16):
            for stride in (1, 2, 3, 8):
                assert sum(rust_count(H, sr, f, stride) for f in range(stride)) == H
                for f in range(stride):
                    assert rust_count(H, sr, f, stride) == shard.strip_row_count(H, sr, f, stride)


def test_crate_files_exist_and_link_the_library():
    toml = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'name = "rm_hip"' in toml and 'build = "build.rs"' in toml and 'links = "rm_hip"' in toml
    assert re.search(r"\[dependencies\]\s*$", toml, re.M)                       # none: nothing to fetch offline
    build = open(os.path.join(CRATE, "build.rs")).read()
    assert "cargo:rustc-link-search=native=" in build and "cargo:rustc-link-lib=dylib=rm_hip" in build
    assert os.path.isdir(os.path.normpath(os.path.join(CRATE, "..", "..", "ray-marching_amd")))   # build.rs's default directory


@pytest.mark.skipif(shutil.which("cargo") is None, reason="no Rust toolchain in this image")
def test_cargo_check():
    from ray_marching_amd import build
    build.build_hip()
    env = dict(os.environ, RM_HIP_LIB_DIR=os.path.dirname(build.HIP_SO))
    p = subprocess.run(["cargo", "check", "--offline", "--tests"], cwd=CRATE, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr
