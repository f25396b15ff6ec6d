"""Properties of the gfx950 code objects that the measured performance rests on, checked on the device assembly hipcc
emits (cross-compiles without a GPU): no register spills in the hot kernels, register budgets that keep the occupancy
DESIGN.md section 4 assumes, and the issue phases of "Two issue ports and wave priority" still in place - a compiler
or source change that silently re-interleaves the instruction classes would cost 15 % of the throughput without
failing any parity test."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _emit_asm(tmp_path_factory, name):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    src = os.path.join(ROOT, "luminair_amd", "csrc", name + ".hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", src,
                        "-o", str(out)], capture_output=True, text=True, cwd=os.path.dirname(src), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    # the kernel translation units (kernels.hip until round 5) as one listing
    return "\n".join(_emit_asm(tmp_path_factory, n) for n in
                     ("kernels_trace", "kernels_fft", "kernels_merkle", "kernels_logup", "kernels_quotient"))


@pytest.fixture(scope="module")
def fft_asm(tmp_path_factory):
    return _emit_asm(tmp_path_factory, "fft_fixed")


def _kernels(asm):
    """name -> (body text, metadata dict)"""
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n"
                         r"\s+\.vgpr_spill_count:\s+(\d+)", asm):
        meta[m.group(1)] = dict(sgpr_spill=int(m.group(2)), vgpr=int(m.group(3)), vgpr_spill=int(m.group(4)))
    bodies = {}
    for m in re.finditer(r"\n(_ZN3lmn\S+):\s*;\s*@", asm):
        end = asm.find("s_endpgm", m.end())
        bodies[m.group(1)] = asm[m.end():end]
    return {k: (bodies.get(k, ""), v) for k, v in meta.items()}


def _find(ks, *parts):
    hits = [k for k in ks if all(p in k for p in parts)]
    assert hits, parts
    return hits


def test_hot_kernels_do_not_spill_and_keep_their_register_budget(device_asm):
    ks = _kernels(device_asm)
    for name in _find(ks, "k_merkle_fused"):
        _, md = ks[name]
        assert md["vgpr_spill"] == 0, (name, md)
        assert md["vgpr"] <= 96, (name, md)          # >= 5 waves per SIMD (512 VGPRs), DESIGN.md section 4
    for part in ("k_fft_staged", "k_fft_interp_extend"):
        for name in _find(ks, part):
            _, md = ks[name]
            assert md["vgpr_spill"] == 0, (name, md)
            assert md["vgpr"] <= 64, (name, md)      # 8 waves per SIMD: these kernels live on latency hiding
    for part in ("k_quotientsILi3E", "k_quotientsILi4E", "k_compositionILi0E", "k_logup_fracs", "k_eval_at_point",
                 "k_transpose_pad"):
        for name in _find(ks, part):
            assert ks[name][1]["vgpr_spill"] == 0, (name, ks[name][1])
    # the one- and two-batch FRI quotient launches run at 8 waves per SIMD (16 waves per SIMD in two whole rounds)
    for part in ("k_quotients_occILi1E", "k_quotients_occILi2E"):
        for name in _find(ks, part):
            md = ks[name][1]
            assert md["vgpr_spill"] == 0 and md["vgpr"] <= 64, (name, md)


def _classes(body):
    """the VALU instruction stream of a kernel as a string: S = first-port-only class, f = plain class, 3 / 0 = s_setprio"""
    port0 = ("v_add3_u32", "v_alignbit_b32", "v_mad_u64_u32", "v_min_u32", "v_max_u32", "v_mul_lo_u32", "v_mul_hi_u32",
             "v_perm_b32", "v_lshl_add_u32", "v_lshl_add_u64", "v_lshlrev_b32", "v_lshl_or_b32", "v_and_or_b32", "v_xad_u32",
             "v_bfe_u32", "v_lshlrev_b64", "v_lshrrev_b64")
    out = []
    for line in body.split("\n"):
        m = re.match(r"\s*(s_setprio|v_[a-z0-9_]+)\s*(\d+)?", line)
        if not m:
            continue
        op = m.group(1)
        if op == "s_setprio":
            out.append(m.group(2))
        elif op.endswith("_dpp") or op.endswith("_sdwa") or any(op.startswith(p) for p in port0):
            out.append("S")
        else:
            out.append("f")
    return "".join(out)


def test_blake2s_half_rounds_are_issued_in_priority_phases(device_asm):
    """k_merkle_fused<1>: at least two whole compressions (40 half rounds) written as class-grouped runs - four add3,
    [prio 0] four xor, [prio 3] four alignbit, [prio 0] eight plain, [prio 3] eight first-port, ..."""
    ks = _kernels(device_asm)
    name, = _find(ks, "k_merkle_fusedILi1ELi16E")
    seq = _classes(ks[name][0])
    half = "SSSS0ffff3SSSS0ffffffff3SSSSSSSS0ffff3SSSS0ffffffff3SSSS"
    assert seq.count(half) >= 40, seq.count(half)


def _count(body, op):
    return len(re.findall(r"^\s*" + op + r"\b", body, re.M))


def test_leaf_compressions_skip_the_first_port_additions_of_zero_message_words(device_asm):
    """blake2s.h b2_compress_fresh_nz: a leaf of NZ < 16 columns (and every FRI layer's 4-word leaf) adds its zero message
    words with two-operand v_add_u32 in the low-priority phase instead of v_add3_u32 on the first issue port.  A fused
    launch holds two leaf and two node compressions as straight-line code: 4 x 160 message additions in the 16-column
    instantiation; 120 of a 4-column leaf's 160 are zero words, 40 of a 12-column leaf's, 10 of a 15-column leaf's."""
    ks = _kernels(device_asm)
    add3 = {}
    for nz in (4, 8, 12, 15, 16):
        name, = _find(ks, "k_merkle_fusedILi1ELi%dE" % nz)
        body, md = ks[name]
        add3[nz] = _count(body, "v_add3_u32")
        assert md["vgpr_spill"] == 0 and md["sgpr_spill"] == 0, (name, md)
        # the phase structure survives: every half round still switches priority around its rotations
        assert _count(body, "s_setprio") >= 600, (name, _count(body, "s_setprio"))
    assert add3[16] >= 640, add3
    assert add3[16] - add3[4] >= 2 * 115, add3      # two leaf compressions x (120 zero-word additions - the folded first half round's)
    assert add3[16] - add3[12] >= 2 * 38, add3
    assert add3[16] - add3[15] >= 2 * 9, add3
    assert add3[4] < add3[8] < add3[12] < add3[15] < add3[16], add3
    fri, = _find(ks, "k_merkle_fusedILi3E")
    assert add3[16] - _count(ks[fri][0], "v_add3_u32") >= 2 * 115, fri


def _addr64(body):
    """v_lshl_add_u64 with an SGPR operand: a 64-bit vector address formed from a uniform base (the other v_lshl_add_u64
    are the 64-bit accumulations of the lazy M31 dot products)"""
    return len(re.findall(r"v_lshl_add_u64 v\[\d+:\d+\], (?:s\[\d+:\d+\], \d+, v\[\d+:\d+\]|v\[\d+:\d+\], \d+, s\[\d+:\d+\])", body))


def test_column_loads_of_the_qm31_and_leaf_kernels_take_uniform_bases_from_sgprs(device_asm):
    """kernels_common.h ld_ub / ld_col: the column loads of k_composition, k_eval_at_point, k_logup_fracs, the FRI quotient
    kernels and the Merkle leaf loads are raw buffer accesses (SGPR resource + one 32-bit lane offset) - no 64-bit vector
    address per load (27 of k_composition<0>'s 862 vector instructions, 32 of k_eval_at_point's 919, one per column and row
    in k_quotients before) - and building the resources does not spill SGPRs in the hot instantiations."""
    ks = _kernels(device_asm)
    for part, max_addr in (("k_compositionILi0E", 0), ("k_compositionILi1E", 0), ("k_eval_at_point", 0), ("k_logup_fracsILi3E", 12),   # (+ the row-major branch of LMN_ROWS_FUSION: per-lane row pointers)
                           ("k_merkle_fusedILi1ELi15E", 6), ("k_merkle_fusedILi1ELi12E", 6), ("k_merkle_fusedILi3E", 6)):
        name, = _find(ks, part)
        body, md = ks[name]
        assert _addr64(body) <= max_addr, (part, _addr64(body))
        assert _count(body, "buffer_load_dword(x2)?") >= 9, (part, _count(body, "buffer_load_dword(x2)?"))
        assert md["sgpr_spill"] == 0 and md["vgpr_spill"] == 0, (part, md)
    # the FRI quotient kernels read their (column, coefficient) table through the scalar cache: no LDS copy, the column
    # loop's loads take their base from SGPRs
    for part in ("k_quotients_occILi1E", "k_quotients_occILi2E"):
        name, = _find(ks, part)
        body, md = ks[name]
        assert _count(body, "ds_read_b(64|128)") == 0 and _count(body, "s_load_dwordx4") >= 6, part
        assert _count(body, "buffer_load_dword") >= 12 and md["vgpr"] <= 64 and md["vgpr_spill"] == 0, (part, md)


def test_butterfly_layers_are_issued_in_priority_phases(device_asm):
    """k_fft_staged<false>: the multiplication phase of a layer of 8 butterflies = 8 v_mad_u64_u32 + 8 v_alignbit_b32 at
    priority 3, then the plain phase, then 8 v_min_u32 at priority 3 (kernels_fft.hip m_mul_phased / radix_butterflies)."""
    ks = _kernels(device_asm)
    for part in ("k_fft_stagedILb0E", "k_fft_stagedILb1E", "k_fft_interp_extend"):
        name, = _find(ks, part)
        body = ks[name][0]
        seq = _classes(body)
        assert seq.count("3") >= 12 and seq.count("0") >= 12, (part, seq.count("3"), seq.count("0"))
        assert re.search(r"3S{16}0f{12,}3S{8}0", seq), part     # one R = 4 layer: 16 first-port, plain run, 8 min
        assert "scratch_" not in body, part


def _valu_per_butterfly(body):
    valu = len(re.findall(r"^\s*v_[a-z0-9_]+", body, re.M))
    mads = len(re.findall(r"^\s*v_mad_u64_u32", body, re.M))      # one multiplication per butterfly
    return valu / mads, mads


def test_fixed_shape_fft_kernels_stay_close_to_11_vector_instructions_per_butterfly(fft_asm):
    """fft_fixed.hip: 11 instructions of arithmetic per butterfly (doubled twiddles) plus at most 1.5 of everything else -
    addressing, LDS indices, twiddle handling, the 2^-n rotation - in the emitted gfx950 code; no spills, and the register
    budgets the measured occupancy rests on (the generic k_fft_staged sits at ~19 per butterfly)."""
    ks = _kernels(fft_asm)
    assert len(_find(ks, "k_fft_fx")) >= 12 and len(_find(ks, "k_fft_interp_extend_fx")) == 4
    for name, (body, md) in ks.items():
        assert md["vgpr_spill"] == 0 and md["sgpr_spill"] == 0 and "scratch_" not in body, (name, md)
        per, mads = _valu_per_butterfly(body)
        hot = any(h in name for h in ("k_fft_fxILb0ELi12E", "interp_extend_fxILi8E", "interp_extend_fxILi9E"))
        assert per <= (12.5 if hot else 15.0), (name, per, mads)   # inverse passes: + the 2^-n rotation; 5 - 6 layer tiles: short stages
        assert md["vgpr"] <= (64 if "interp_extend" in name else 80), (name, md)
        seq = _classes(body)
        assert seq.count("3") >= 6 and seq.count("0") >= 6, name        # the issue phases are in the emitted code
    # no 64-bit per-lane address arithmetic in the strided stages: buffer addressing with scalar offsets
    body = ks[_find(ks, "k_fft_interp_extend_fxILi8E")[0]][0]
    assert "buffer_store_dword" in body and "v_addc_co_u32" not in body


def test_merkle_kernel_has_no_constant_moves_in_front_of_its_compressions(device_asm):
    """k_merkle_fused<1>: the initial hash state enters the first half round as literal operands (blake2s.h b2_half_first)
    and the zero message words are set once per kernel - at most ~25 v_mov per compression site remain (round 3: 46,
    16 of them state constants and 16 zeroed message words in front of every leaf)."""
    ks = _kernels(device_asm)
    name, = _find(ks, "k_merkle_fusedILi1ELi16E")
    body = ks[name][0]
    movs = len(re.findall(r"^\s*v_mov_b32", body, re.M))
    sites = len(re.findall(r"^\s*v_alignbit_b32", body, re.M)) / 320.0      # 320 rotations per compression
    assert sites >= 4 and movs / sites <= 26, (movs, sites)
    assert re.search(r"v_add_u32 v\d+, 0x[0-9a-f]+, v\d+\n\s*v_add_u32 v\d+, 0x[0-9a-f]+, v\d+", body) or "0x510e527f" in body.lower()
