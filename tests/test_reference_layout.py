"""The boundary data of the hot path - table kinds, row field order, column counts, relation counts, padding rows,
claim / component order, relation draw order - checked BY MACHINE against tests/golden/reference_layout.json, which
tests/golden/extract_reference_layout.py parsed out of the reference's Rust sources (crates/air/src/pie.rs:31-66, every
components/**/table.rs, components/mod.rs:261-601, lookups/mod.rs:18-51).  For the 15 components the reference's
known-answer proof does not exercise, this is the one piece of reference-held truth there is."""
import ctypes as C
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def layout():
    with open(os.path.join(ROOT, "tests", "golden", "reference_layout.json")) as f:
        return json.load(f)


def test_fixture_is_internally_consistent(layout):
    assert [k["kind"] for k in layout["kinds"]] == list(range(17))
    for k in layout["kinds"]:
        camel = ["".join(p.capitalize() for p in f.split("_")) for f in k["row_fields"]]
        assert k["column_variants"] == camel, k["variant"]                      # Column enum = row struct, same order
        assert [k["column_index"][v] for v in camel] == list(range(len(camel))), k["variant"]   # index() is the position
        assert k["count"][0] == k["n_trace_columns"] == len(k["row_fields"]), k["variant"]
    assert layout["claim_fields"] == layout["component_order"]                  # claims are mixed in component order
    assert layout["claim_fields"] == [re.sub(r"(?<!^)(?=[A-Z])", "_", k["variant"]).lower() for k in layout["kinds"]]
    assert layout["interaction_elements_draw_order"] == ["node_elements", "lookup_elements"]
    assert layout["lookup_elements_draw_order"] == layout["lookup_fields"] == ["sin", "exp2", "log2", "range_check"]


def test_fixture_is_current_when_the_reference_is_present(layout, tmp_path):
    """in the build container the extractor is re-run: a stale fixture fails here"""
    if not os.path.isdir("/root/reference/crates/air/src"):
        pytest.skip("reference sources not on this box")
    import importlib.util
    spec = importlib.util.spec_from_file_location("extract", os.path.join(ROOT, "tests", "golden", "extract_reference_layout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.OUT = str(tmp_path / "layout.json")
    mod.main()
    assert json.load(open(mod.OUT)) == layout


def test_c_abi_matches_the_reference_layout(layout, hip_lib_path):
    from luminair_amd import backend
    lib = backend.Library(hip_lib_path).lib
    for k in layout["kinds"]:
        n = len(k["row_fields"])
        assert lib.lmn_kind_columns(k["kind"]) == n, k["variant"]
        assert lib.lmn_kind_relations(k["kind"]) == k["count"][1], k["variant"]
        row = (C.c_uint32 * n)()
        assert lib.lmn_kind_padding_row(k["kind"], row) == 0
        want = [k["padding_nonzero"].get(f, 0) for f in k["row_fields"]]
        assert list(row) == want, k["variant"]
    assert lib.lmn_kind_columns(17) == 0


def test_python_mirror_matches_the_reference_layout(layout):
    from luminair_amd import pie
    assert [m.name for m in pie.TraceTableKind] == [k["variant"] for k in layout["kinds"]]
    assert [int(m) for m in pie.TraceTableKind] == [k["kind"] for k in layout["kinds"]]
    assert {int(kk): v for kk, v in pie.N_COLUMNS.items()} == {k["kind"]: len(k["row_fields"]) for k in layout["kinds"]}
    assert list(pie._CLAIM_FIELDS) == layout["claim_fields"]
    assert list(pie._LOOKUP_FIELDS) == layout["lookup_fields"]


def test_oracle_components_match_the_reference_layout(layout):
    from oracle import air
    assert sorted(air.COMPONENTS) == list(range(17))
    for k in layout["kinds"]:
        comp = air.COMPONENTS[k["kind"]]
        assert comp.n_cols == len(k["row_fields"]), k["variant"]
        assert len(comp.relations) == k["count"][1], k["variant"]
        assert [int(v) for v in comp.padding] == [k["padding_nonzero"].get(f, 0) for f in k["row_fields"]], k["variant"]


def test_integration_md_flatten_arms_match_the_reference_layout(layout):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    arms = re.findall(r"TraceTable::(\w+) \{ table \} => flat!\((\d+), table, \[([^\]]*)\]\)", text)
    assert len(arms) == 17
    for (variant, kind, fields), k in zip(arms, layout["kinds"]):
        assert (variant, int(kind)) == (k["variant"], k["kind"])
        assert [f.strip() for f in fields.split(",")] == k["row_fields"], variant
