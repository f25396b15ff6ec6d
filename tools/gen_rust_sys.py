#!/usr/bin/env python3
"""Generates bindings/rust/luminair-hip-sys/src/lib.rs - the `extern "C"` side of a Rust binding - from
include/luminair_hip.h and include/luminair_hip_batch.h.

No Rust toolchain exists in this image, so the crate cannot be compiled here; what CAN be done
without one is to derive the declarations mechanically from the header a C compiler does check, and to test the
derivation (tests/test_rust_bindings.py: every exported symbol is declared with the header's arity, every `#[repr(C)]`
struct has the size and field offsets `ctypes` computes for the same header).  First contact with a toolchain is then
`cargo build` of generated code, and INTEGRATION.md's hand-written shim sits on top of it.

Usage: gen_rust_sys.py [--check]     (--check: exit 1 if the committed file differs from what would be generated)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", "luminair_hip.h"), os.path.join(ROOT, "include", "luminair_hip_batch.h")]
OUT = os.path.join(ROOT, "bindings", "rust", "luminair-hip-sys", "src", "lib.rs")

SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16",
           "int32_t": "i32", "int64_t": "i64", "int": "c_int", "unsigned": "c_uint", "size_t": "usize", "float": "f32",
           "double": "f64", "char": "c_char", "void": "c_void"}
SIZES = {"u8": 1, "i8": 1, "c_char": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "c_int": 4, "c_uint": 4, "f32": 4, "u64": 8,
         "i64": 8, "usize": 8, "f64": 8}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def rust_type(ctype, opaque, structs):
    """C type text (no declarator name) -> Rust type"""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split()
    base, const_base, ptrs = None, False, []      # ptrs: list of constness of what each '*' points to, innermost first
    pending_const = False
    for tok in toks:
        if tok == "const":
            pending_const = True
        elif tok == "struct":
            continue
        elif tok == "*":
            ptrs.append(pending_const if base is not None and not ptrs else pending_const)
            pending_const = False
        else:
            base = tok
            const_base = pending_const
            pending_const = False
    if base in SCALARS:
        r = SCALARS[base]
    elif base in opaque or base in structs:
        r = base
    else:
        raise ValueError("unknown C type %r" % ctype)
    # walk pointers from the innermost: pointee constness of the first '*' is const_base, of later ones what preceded them
    consts = [const_base] + ptrs[1:] if ptrs else []
    for k in range(len(ptrs)):
        r = ("*const " if consts[k] else "*mut ") + r
    return r


def parse(headers):
    defines, structs, opaque, funcs, fnptr_structs = [], {}, [], [], {}
    for h in headers:
        raw = open(h).read()
        for m in re.finditer(r"^#define\s+(LMN_[A-Z0-9_]+)\s+(\(?-?[0-9][0-9a-fx]*u?\)?|\([A-Z_ |()]+\)|LMN_[A-Z_]+)\s*(?:/\*.*)?$", raw, re.M):
            defines.append((m.group(1), m.group(2)))
        text = strip_comments(raw)
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
            if m.group(2) not in opaque:
                opaque.append(m.group(2))
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, re.S):
            name, body = m.group(3), m.group(2)
            fields = []
            for decl in [d.strip() for d in body.split(";") if d.strip()]:
                fp = re.match(r"(.+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", decl, re.S)
                if fp:  # function pointer field
                    fields.append((fp.group(2), ("fnptr", fp.group(1).strip(), fp.group(3))))
                    continue
                # type = the leading type words; the rest is a comma-separated list of declarators (`*name`, `name[N]`)
                toks = decl.replace("*", " * ").split()
                k = 0
                while k < len(toks) and (toks[k] in ("const", "struct", "unsigned") or toks[k] in SCALARS or toks[k] in structs
                                         or toks[k] in opaque):
                    k += 1
                ctype = " ".join(toks[:k])
                dmap = {a: b for a, b in defines}
                for item in " ".join(toks[k:]).split(","):
                    item = item.replace(" ", "")
                    stars = item.count("*")
                    im = re.match(r"\**(\w+)((?:\[\w+\])*)$", item)
                    if not im:
                        raise ValueError("cannot parse field %r of %s" % (decl, name))
                    dims = [int(d) if d.isdigit() else int(dmap[d].strip("()u")) for d in re.findall(r"\[(\w+)\]", im.group(2))]
                    fields.append((im.group(1), ("data", ctype + " *" * stars, dims)))
            structs[name] = fields
        body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        body = re.sub(r"#.*", " ", body)                      # preprocessor lines
        body = body.replace('extern "C" {', " ").replace("}", " ")
        for stmt in body.split(";"):
            m = re.match(r"\s*((?:const\s+)?\w+(?:\s*\*+)?)\s*(lmn_\w+)\s*\((.*)\)\s*$", stmt, re.S)
            if m:
                funcs.append((m.group(2), m.group(1).strip(), m.group(3).strip()))
    opaque = [o for o in opaque if o not in structs]
    return defines, structs, opaque, funcs


def params(text, opaque, structs):
    out = []
    if text.strip() in ("", "void"):
        return out
    depth, cur, parts = 0, "", []
    for ch in text:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    for i, p in enumerate(parts):
        p = " ".join(p.split())
        fp = re.match(r"(.+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", p)
        if fp:
            out.append((fp.group(2), fnptr_type(fp.group(1), fp.group(3), opaque, structs)))
            continue
        m = re.match(r"(.+?)\b(\w+)((?:\[\w*\])*)$", p)
        ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
        if name in SCALARS or name in opaque or name in structs or ctype in ("const", "unsigned", ""):   # unnamed parameter
            ctype, name = p, "arg%d" % i
            arr = ""
        if arr:
            ctype += " *"     # an array parameter is a pointer
        out.append((name, rust_type(ctype, opaque, structs)))
    return out


def fnptr_type(ret, args, opaque, structs):
    ps = params(args, opaque, structs)
    r = "" if ret.strip() == "void" else " -> " + rust_type(ret, opaque, structs)
    return "Option<unsafe extern \"C\" fn(%s)%s>" % (", ".join("%s: %s" % (n, t) for n, t in ps), r)


def field_type(f, opaque, structs):
    kind = f[0]
    if kind == "fnptr":
        return fnptr_type(f[1], f[2], opaque, structs)
    t = rust_type(f[1], opaque, structs)
    for d in reversed(f[2]):
        t = "[%s; %d]" % (t, d)
    return t


def layout(structs, opaque):
    """name -> (size, align, [(field, offset, size)]) under repr(C) on x86-64 / LP64"""
    done = {}

    def of(t):
        if t.startswith("*") or t.startswith("Option<"):
            return 8, 8
        m = re.match(r"\[(.+); (\d+)\]$", t)
        if m:
            s, a = of(m.group(1))
            return s * int(m.group(2)), a
        if t in SIZES:
            return SIZES[t], SIZES[t]
        if t in done:
            return done[t][0], done[t][1]
        return one(t)[:2]

    def one(name):
        off, align, out = 0, 1, []
        for fname, f in structs[name]:
            s, a = of(field_type(f, opaque, structs))
            off = (off + a - 1) // a * a
            out.append((fname, off, s))
            off += s
            align = max(align, a)
        size = (off + align - 1) // align * align
        done[name] = (size, align, out)
        return done[name]
    for n in structs:
        if n not in done:
            one(n)
    return done


def generate():
    defines, structs, opaque, funcs = parse(HEADERS)
    L = ["// GENERATED by tools/gen_rust_sys.py from include/luminair_hip.h and include/luminair_hip_batch.h - do not edit.",
         "// `extern \"C\"` declarations of libluminair_hip.so / libluminair_hip_batch.so for a Rust binding of the reference",
         "// (/root/reference/crates/prover/src/prover.rs:28-31 `prove`, crates/verifiers/rust/src/verifier.rs:21 `verify`; the",
         "// safe shim on top of these is INTEGRATION.md's `crates/prover/src/hip.rs`).  Never compiled in the image that made it",
         "// (no Rust toolchain there): tests/test_rust_bindings.py checks it against the header and the built library instead.",
         "#![allow(non_camel_case_types, non_upper_case_globals, non_snake_case, dead_code)]",
         "use core::ffi::{c_char, c_int, c_uint, c_void};", ""]
    seen = set()
    for name, val in defines:
        if name in seen or name in ("LMN_API_VERSION",) and False:
            continue
        seen.add(name)
        v = val.strip()
        if re.fullmatch(r"\(?-?[0-9][0-9a-fx]*u?\)?", v):
            lit = v.strip("()")
            if lit.startswith("-") or name == "LMN_OK":
                L.append("pub const %s: c_int = %s;" % (name, lit))
            else:
                L.append("pub const %s: u32 = %s;" % (name, lit.rstrip("u")))
        else:
            expr = v.strip("()") if v.startswith("(") else v
            L.append("pub const %s: u32 = %s;" % (name, expr))
    L.append("")
    for o in opaque:
        L += ["#[repr(C)]", "pub struct %s {" % o, "    _private: [u8; 0],", "}", ""]
    lay = layout(structs, opaque)
    for name, fields in structs.items():
        L += ["#[repr(C)]", "#[derive(Clone, Copy)]", "pub struct %s {" % name]
        for fname, f in fields:
            rn = "r#type" if fname == "type" else fname
            L.append("    pub %s: %s," % (rn, field_type(f, opaque, structs)))
        L += ["}", "// size %d, align %d" % (lay[name][0], lay[name][1]), ""]
    L.append("extern \"C\" {")
    for name, ret, args in funcs:
        ps = params(args, opaque, structs)
        r = "" if ret == "void" else " -> " + rust_type(ret, opaque, structs)
        L.append("    pub fn %s(%s)%s;" % (name, ", ".join("%s: %s" % (n, t) for n, t in ps), r))
    L += ["}", ""]
    return "\n".join(L), structs, funcs, lay


def main():
    text, _, _, _ = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
