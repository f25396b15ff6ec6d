#!/usr/bin/env python3
"""Extracts the BOUNDARY DATA of the prove hot path from the reference's Rust sources into a fixture:
tests/golden/reference_layout.json.  Run in the build container only (/root/reference does not exist on the GPU box):

    python tests/golden/extract_reference_layout.py

What it reads (and nothing else - no Rust text is copied, only names, orders and small integers):
  * crates/air/src/pie.rs                      `enum TraceTable` variant order = the `kind` numbers of lmn_table
  * crates/air/src/components/**/table.rs      `*TraceTableRow` field order, `padding()` cells that are not zero,
                                               the `*Column::index()` arms, `TraceColumn::count()`
  * crates/air/src/components/**/witness.rs    `N_TRACE_COLUMNS`
  * crates/air/src/lib.rs                      `LuminairClaim` field order (Fiat-Shamir mixing order of the claims)
  * crates/air/src/components/mod.rs           component construction order in `LuminairComponents::new`,
                                               `LuminairInteractionElements::draw`
  * crates/air/src/components/lookups/mod.rs   `Lookups` field order and `LookupElements::draw` order
tests/test_reference_layout.py asserts the C ABI (`lmn_kind_columns`, `lmn_kind_relations`, `lmn_kind_padding_row`),
luminair_amd/pie.py, oracle/air.py and the `flat!` lists of INTEGRATION.md against the fixture."""
import json
import os
import re
import sys

REF = "/root/reference/crates/air/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_layout.json")
DEFAULT_FP_SCALE = 12      # numerair `DEFAULT_FP_SCALE` (un-vendored; the KAT pins 2^12)


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def snake(name):
    return re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()


def block(text, start_pat):
    """text of the brace block that follows the first match of start_pat"""
    m = re.search(start_pat, text)
    if not m:
        return None
    i = text.index("{", m.end() - 1)
    depth, j = 0, i
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j]
        j += 1


def m31_value(expr):
    expr = expr.strip()
    if expr in ("M31::zero()", "M31::from(0)", "BaseField::zero()"):
        return 0
    if expr in ("M31::one()", "M31::from(1)", "BaseField::one()"):
        return 1
    m = re.fullmatch(r"M31::from(?:_u32_unchecked)?\((.+)\)", expr)
    if m:
        return int(eval(m.group(1).replace("DEFAULT_FP_SCALE", str(DEFAULT_FP_SCALE)), {"__builtins__": {}}))
    raise ValueError("padding expression not understood: %r" % expr)


def table_module(variant):
    """TraceTable variant -> module path under components/ (SinLookup -> lookups/sin, RangeCheckLookup -> lookups/range_check)"""
    if variant.endswith("Lookup"):
        return "lookups/" + snake(variant[:-len("Lookup")])
    return snake(variant)


def main():
    pie = read("pie.rs")
    variants = re.findall(r"^\s*([A-Z]\w*)\s*\{\s*table:\s*(\w+)\s*\}", block(pie, r"pub enum TraceTable\s*\{"), re.M)
    kinds = []
    for kind, (variant, table_type) in enumerate(variants):
        mod = table_module(variant)
        t = read("components/%s/table.rs" % mod)
        w = read("components/%s/witness.rs" % mod)
        row_type = re.search(r"pub table:\s*Vec<(\w+)>", block(t, r"pub struct %s\s*\{" % table_type)).group(1)
        fields = re.findall(r"pub (\w+):\s*(?:M31|BaseField)", block(t, r"pub struct %s\s*\{" % row_type))
        pad_body = block(block(t, r"fn padding\([^)]*\)\s*->\s*Self\s*\{"), r"Self\s*\{")
        pad = {}
        for name, expr in re.findall(r"(\w+):\s*([^,\n]+(?:\([^)]*\))?),", pad_body):
            v = m31_value(expr)
            if v:
                pad[name] = v
        col_enum = re.search(r"pub enum (\w+Column)\s*\{", t).group(1)
        enum_body = re.sub(r"//[^\n]*", "", block(t, r"pub enum %s\s*\{" % col_enum))
        col_variants = [v.strip() for v in enum_body.split(",") if v.strip()]
        arms = {v: int(i) for v, i in re.findall(r"Self::(\w+)\s*=>\s*(\d+)", block(t, r"fn index\(self\)\s*->\s*usize\s*\{"))}
        n_const = int(re.search(r"const N_TRACE_COLUMNS:\s*usize\s*=\s*(\d+)", w).group(1))
        cnt = re.search(r"fn count\(\)\s*->\s*\(usize,\s*usize\)\s*\{\s*\((\w+),\s*(\d+)\)", t)
        n_cols = n_const if cnt.group(1) == "N_TRACE_COLUMNS" else int(cnt.group(1))
        kinds.append({"kind": kind, "variant": variant, "module": "components/" + mod, "row_fields": fields,
                      "padding_nonzero": pad, "column_enum": col_enum, "column_variants": col_variants, "column_index": arms,
                      "n_trace_columns": n_const, "count": [n_cols, int(cnt.group(2))]})
    lib = read("lib.rs")
    claim_fields = re.findall(r"pub (\w+):\s*Option<", block(lib, r"pub struct LuminairClaim\s*\{"))
    comps = read("components/mod.rs")
    new_body = block(comps, r"impl LuminairComponents\s*\{")
    component_order = re.findall(r"^\s*let (\w+) =\s*if let Some\(ref \w+\) = claim\.\w+", new_body, re.M)
    elem_draw = re.findall(r"let (\w+) = \w+::draw\(channel\)", block(comps, r"impl LuminairInteractionElements\s*\{"))
    lk = read("components/lookups/mod.rs")
    lookup_fields = re.findall(r"pub (\w+):\s*Option<", block(lk, r"pub struct Lookups\s*\{"))
    lookup_draw = re.findall(r"(\w+):\s*\w+::draw\(channel\)", block(lk, r"impl LookupElements\s*\{"))
    out = {"source": "gizatechxyz/LuminAIR crates/air/src (names, orders and small integers only)",
           "generated_by": "tests/golden/extract_reference_layout.py", "kinds": kinds, "claim_fields": claim_fields,
           "component_order": component_order, "interaction_elements_draw_order": elem_draw,
           "lookup_fields": lookup_fields, "lookup_elements_draw_order": lookup_draw}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT, ":", len(kinds), "kinds;", len(component_order), "components")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("the reference sources are not available here (%s)" % REF)
    main()
