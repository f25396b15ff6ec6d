"""bindings/rust/luminair-hip-sys (what can be had of the Rust side without a Rust toolchain): the
generated `extern "C"` declarations against the headers, the exported symbols and the C struct layouts."""
import ctypes as C
import importlib.util
import os
import re

from luminair_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
LIB_RS = os.path.join(ROOT, "bindings", "rust", "luminair-hip-sys", "src", "lib.rs")


def test_committed_bindings_are_what_the_generator_makes_of_the_headers():
    text, _, _, _ = gen.generate()
    assert open(LIB_RS).read() == text, "run tools/gen_rust_sys.py"


def test_every_exported_symbol_is_declared_with_the_headers_arity():
    text, structs, funcs, _ = gen.generate()
    declared = {}
    for m in re.finditer(r"pub fn (lmn_\w+)\((.*?)\)(?: -> [^;]+)?;", text):
        args = m.group(2).strip()
        depth, n = 0, (1 if args else 0)
        for ch in args:
            depth += ch in "(<"
            depth -= ch in ")>"
            n += ch == "," and depth == 0
        declared[m.group(1)] = n
    batch = ["lmn_batch_create", "lmn_batch_prove", "lmn_batch_destroy", "lmn_batch_last_error", "lmn_batch_counter"]
    assert set(declared) == set(backend.EXPORTS) | set(batch)
    # arity as the C prototypes have it
    header = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "luminair_hip.h")).read()
                    + open(os.path.join(ROOT, "include", "luminair_hip_batch.h")).read(), flags=re.S)
    for name, n in declared.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, header, re.S)
        assert m, name
        a = m.group(1).strip()
        want = 0 if a in ("", "void") else a.count(",") + 1 - sum(seg.count(",") for seg in re.findall(r"\([^()]*\)", a)[1:])
        if "(*" not in a:
            assert n == want, (name, n, want)
    # what the libraries really export (hipcc cross-compiles here; the product library loads without a GPU)
    for path in (backend.DEFAULT_LIB, os.path.join(os.path.dirname(backend.DEFAULT_LIB), "libluminair_hip_batch.so")):
        if os.path.exists(path):
            lib = C.CDLL(path)
            for name in declared:
                if name.startswith("lmn_batch_") and not path.endswith("_batch.so"):
                    continue
                getattr(lib, name)


def test_repr_c_layouts_equal_the_ctypes_layouts_of_the_python_binding():
    _, structs, _, lay = gen.generate()
    pairs = {"lmn_config": backend.LmnConfig, "lmn_table": backend.LmnTable, "lmn_lut": backend.LmnLut,
             "lmn_settings": backend.LmnSettings, "lmn_range": backend.LmnRange, "lmn_timings": backend.LmnTimings,
             "lmn_node_info": backend.LmnNodeInfo, "lmn_view": backend.LmnView, "lmn_collective": backend.LmnCollective,
             "lmn_transcript_step": backend.LmnTranscriptStep, "lmn_verify_report": backend.LmnVerifyReport}
    assert set(pairs) == set(structs), set(structs) ^ set(pairs)
    for name, ct in pairs.items():
        size, _, fields = lay[name]
        assert size == C.sizeof(ct), (name, size, C.sizeof(ct))
        assert [f for f, _, _ in fields] == [f[0] for f in ct._fields_], name
        for fname, off, fsize in fields:
            d = getattr(ct, fname)
            assert (off, fsize) == (d.offset, d.size), (name, fname, off, fsize, d.offset, d.size)
