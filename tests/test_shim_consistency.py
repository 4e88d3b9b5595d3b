"""CPU: the Rust shim (shim/src/hip.rs + shim/patches/*.diff) against the C header.  No Rust toolchain exists in this image, so what can
be checked mechanically is checked here: every `extern "C"` declaration of the shim has the name, arity, argument order, pointer depth,
constness and base type of its twin in include/sbn254.h (and vice versa), every ABI function the shim's bodies call is declared, every
`crate::hip::` item the patches use exists, nothing is left as a placeholder, and the committed patches are exactly what the recorded
edits produce on the reference's files (checked where the reference lies; skipped on the GPU box)."""
import os
import re
import subprocess
import sys

import pytest
from conftest import ROOT

HIP_RS = os.path.join(ROOT, "shim", "src", "hip.rs")
HEADER = os.path.join(ROOT, "include", "sbn254.h")
PATCHES = os.path.join(ROOT, "shim", "patches")

C_BASE = {"int": "c_int", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "double": "f64", "char": "c_char", "void": "c_void"}


def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def c_decls():
    """name -> (ret, [argtype...]) with each type normalised to (base, [constness of each pointer level's pointee, outermost last])"""
    out = {}
    src = _strip_c(open(HEADER).read())
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(sbn_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if not ret or ret.startswith("typedef"):
            continue
        lst = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                is_arr = bool(re.search(r"\[\s*\d*\s*\]$", a))
                a = re.sub(r"\[\s*\d*\s*\]$", "", a).strip()
                ctype = re.match(r"(.*?)([A-Za-z_]\w*)$", a).group(1).strip() + ("*" if is_arr else "")
                lst.append(norm_c(ctype))
        out[name] = (norm_c(ret) if ret != "void" else None, lst)
    return out


def norm_c(ctype):
    toks = ctype.replace("*", " * ").split()
    i = toks.index("*") if "*" in toks else len(toks)
    base = [t for t in toks[:i] if t != "const"]
    assert len(base) == 1, ctype
    levels = []
    pointee_const = "const" in toks[:i]
    rest = toks[i:]
    k = 0
    while k < len(rest):
        levels.append(pointee_const)
        pointee_const = k + 1 < len(rest) and rest[k + 1] == "const"
        k += 2 if pointee_const else 1
    return (C_BASE.get(base[0], base[0]), levels)


def norm_rust(rtype):
    levels = []
    t = rtype.strip()
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+(.*)$", t)
        levels.append(m.group(1) == "const")
        t = m.group(2).strip()
    return (t, list(reversed(levels)))


def rust_decls():
    src = open(HIP_RS).read()
    blk = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (sbn_[a-z0-9_]+)\((.*?)\)\s*(?:->\s*([^;]+))?;", blk, flags=re.S):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        lst = []
        if args:
            for a in args.split(","):
                lst.append(norm_rust(a.split(":", 1)[1]))
        out[name] = (norm_rust(ret) if ret else None, lst)
    return out


def test_every_declaration_matches_the_header():
    c, r = c_decls(), rust_decls()
    assert len(c) >= 90
    assert sorted(c) == sorted(r), f"only in the header: {sorted(set(c) - set(r))}; only in the shim: {sorted(set(r) - set(c))}"
    for name in c:
        cret, cargs = c[name]
        rret, rargs = r[name]
        assert len(cargs) == len(rargs), f"{name}: {len(cargs)} arguments in the header, {len(rargs)} in the shim"
        for k, (ca, ra) in enumerate(zip(cargs, rargs)):
            # the levels list pointee-constness from the base outwards on the C side; reversed Rust nesting gives the same order
            assert ca[0] == ra[0] and ca[1] == ra[1], f"{name}: argument {k} is {ca} in the header and {ra} in the shim"
        assert cret == rret, f"{name}: return type {cret} vs {rret}"


def test_generated_block_is_current():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_shim_is_complete_on_paper():
    src = open(HIP_RS).read()
    body = src[src.index("// GENERATED-END"):]
    code = re.sub(r"//[^\n]*", "", body)
    for bad in ("todo!", "unimplemented!", "TODO", "FIXME", "..."):
        assert bad not in code, f"placeholder left in the shim: {bad}"
    # every ABI function the bodies call is declared
    declared = set(rust_decls())
    called = set(re.findall(r"\b(sbn_[a-z0-9_]+)\s*\(", code)) - {"sbn_ctx", "sbn_bases", "sbn_table", "sbn_sumcheck", "sbn_bullet", "sbn_group", "sbn_group_bases"}
    assert called <= declared, f"called but not declared: {sorted(called - declared)}"
    assert len(called) >= 25
    # brackets balance (a cheap guard against truncated edits)
    whole = re.sub(r"//[^\n]*", "", src)
    for o, c_ in ("{}", "()", "[]"):
        assert whole.count(o) == whole.count(c_), f"unbalanced {o}{c_}"
    # the items VERDICT r3 found undefined exist now
    for item in ("pub const MIN_GPU_MSM", "pub struct Bases", "pub struct GroupBases", "impl Drop for Bases", "impl Drop for GroupBases", "unsafe impl Send for Bases",
                 "unsafe impl Sync for Bases", "pub fn check_group", "pub fn scalars_canonical", "pub fn sc(", "pub fn prove_cubic(", "pub fn prove_cubic_batched(",
                 "pub struct R1csRounds", "pub struct QuadRounds", "pub fn bullet_prove("):
        assert item in src, item
    assert "bytemuck" not in src                       # not a dependency of the reference (Cargo.toml:7-31)


def test_patches_only_use_what_the_shim_defines():
    src = open(HIP_RS).read()
    defined = set(re.findall(r"pub (?:fn|struct|const|type) ([A-Za-z_][A-Za-z0-9_]*)", src))
    methods = set(re.findall(r"pub fn ([a-z_][a-z0-9_]*)\(", src))
    used = set()
    n = 0
    for f in sorted(os.listdir(PATCHES)):
        txt = open(os.path.join(PATCHES, f)).read()
        n += 1
        added = "\n".join(l[1:] for l in txt.splitlines() if l.startswith("+") and not l.startswith("+++"))
        used |= set(re.findall(r"crate::hip::([A-Za-z_][A-Za-z0-9_]*)", added))
        for o, c_ in ("{}", "()"):
            removed = "\n".join(l[1:] for l in txt.splitlines() if l.startswith("-") and not l.startswith("---"))
            assert added.count(o) - added.count(c_) == removed.count(o) - removed.count(c_), f"{f}: the edit changes the {o}{c_} balance"
        for meth in re.findall(r"hip_rounds\.([a-z_]+)\(", added):
            assert meth in methods, f"{f}: hip_rounds.{meth} is not defined in the shim"
        assert "..." not in added
    assert n == 9
    assert used and used <= defined, f"patches use undefined items: {sorted(used - defined)}"
    assert {"msm_affine", "msm_projective", "commit_row", "commit_rows", "prove_cubic", "prove_cubic_batched", "R1csRounds", "QuadRounds", "bullet_prove", "GensDev", "MIN_GPU_MSM"} <= used
    sc = open(os.path.join(PATCHES, "0003-scalar-repr-transparent.diff")).read()
    assert "+#[repr(transparent)]" in sc


def test_patches_are_what_the_edits_produce_on_the_reference():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("the reference is not on this machine (GPU box)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_shim_patches.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
