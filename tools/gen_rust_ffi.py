#!/usr/bin/env python3
"""include/sbn254.h -> the `extern "C"` block of shim/src/hip.rs (the Rust FFI declarations of every entry point).
    python tools/gen_rust_ffi.py            prints the block
    python tools/gen_rust_ffi.py --check    exit 1 unless shim/src/hip.rs carries exactly this block between its GENERATED markers
tests/test_shim_consistency.py parses both files independently of this generator (name, arity, argument order, pointer-ness)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPAQUE = ["sbn_ctx", "sbn_bases", "sbn_table", "sbn_sumcheck", "sbn_bullet", "sbn_group", "sbn_group_bases"]
SCALAR = {"int": "c_int", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "double": "f64", "char": "c_char", "void": "c_void"}


def c_functions(header_text):
    """[(ret, name, [(ctype, argname)])] for every sbn_* function declared in the header"""
    src = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = []
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(sbn_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef") or not ret:
            continue
        lst = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.search(r"\[\s*\d*\s*\]$", a)
                if arr:
                    a = a[:arr.start()].strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                ctype, an = mm.group(1).strip(), mm.group(2)
                if arr:
                    ctype += "*"
                lst.append((ctype, an))
        out.append((ret, name, lst))
    return out


def rust_type(ctype):
    """C type -> Rust type (pointers read right to left; `const` binds to what precedes the `*`)"""
    t = ctype.replace("*", " * ").split()
    # base: tokens up to the first '*'
    i = t.index("*") if "*" in t else len(t)
    base = [x for x in t[:i] if x not in ("const", "struct")]
    const_base = "const" in t[:i]
    assert len(base) == 1, ctype
    b = base[0]
    r = SCALAR.get(b, b)
    assert b in SCALAR or b in OPAQUE, ctype
    rest = t[i:]
    # walk the pointer levels: each '*' optionally followed by 'const' (constness of that pointer = of the level below the next '*')
    pointee_const = const_base
    k = 0
    while k < len(rest):
        assert rest[k] == "*"
        r = ("*const " if pointee_const else "*mut ") + r
        pointee_const = k + 1 < len(rest) and rest[k + 1] == "const"
        k += 2 if pointee_const else 1
    return r


def rust_ret(ret):
    if ret == "void":
        return ""
    return " -> " + rust_type(ret)


RESERVED = {"r": "r", "st": "st", "in": "input", "type": "ty", "ref": "rf", "mod": "md", "fn": "f", "box": "bx"}


def rust_block(funcs):
    lines = []
    for ret, name, args in funcs:
        a = ", ".join(f"{RESERVED.get(an, an).lower()}: {rust_type(ct)}" for ct, an in args)
        lines.append(f"    pub fn {name}({a}){rust_ret(ret)};")
    return "\n".join(lines)


def main():
    funcs = c_functions(open(os.path.join(ROOT, "include", "sbn254.h")).read())
    block = rust_block(funcs)
    if "--check" in sys.argv:
        src = open(os.path.join(ROOT, "shim", "src", "hip.rs")).read()
        m = re.search(r"// GENERATED-BEGIN[^\n]*\n(.*?)\n\s*// GENERATED-END", src, flags=re.S)
        if not m or m.group(1).rstrip() != block.rstrip():
            print("shim/src/hip.rs: the extern block is stale; run tools/gen_rust_ffi.py"); sys.exit(1)
        print(f"{len(funcs)} declarations match"); return
    print(block)


if __name__ == "__main__":
    main()
