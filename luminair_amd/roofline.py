"""The `roofline` objects of bench.py: one hot kernel family of the prover against the ceilings that bound it.

`frac` (the contract field) = ALGORITHMIC bytes per launch (SURVEY.md §8(d)'s per-pass formulas x the units a launch
processes, counted by the library itself: lmn_timings) / HIP-event launch time / HBM peak.  For the Blake2s tree kernel
that formula counts every node as written and re-read, while the fused launches keep 7/8 of a tree's nodes in registers:
the kernel moves far fewer bytes than the formula says, so next to `frac` the line carries

* `frac_by_counter_traffic`: HBM bytes the rocprofv3 PMC passes measured per launch / launch time / HBM peak - the
  fraction of the memory system the kernel really uses;
* `alu_ceiling`: operations per second against (a) the ceiling measured with both VALU issue ports in use
  (tools/microbench_reconcile.hip) and (b) the architectural bound - the minimal vector-instruction count per operation
  at 0.5 wave-instructions per clock per SIMD (two issue ports, one wave64 instruction per 4 clocks each) on
  1024 SIMDs at the 2.4 GHz maximum clock;
* `bound`: the ceiling the kernel sits closest to.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak
# ALU ceilings measured on MI355X with both VALU issue ports in use (tools/microbench_reconcile.hip `prio` / `bfly`,
# profiles/ceilings/valu_coissue_two_ports.txt)
BLAKE2S_PEAK_GCOMP = 63.8    # G compressions/s chip-wide: 1024 SIMDs x 2.235 GHz x 0.425 instr/clk x 64 lanes / 976 instr
BUTTERFLY_PEAK_G = 4800.0    # G M31 butterflies/s chip-wide: 0.035 butterflies/clk/SIMD x 2.08 GHz x 1024 SIMDs x 64 lanes
# architectural issue bound: 256 CUs x 4 SIMDs, 64 lanes, 2 ports x 1 wave-instruction / 4 clk, 2.4 GHz
N_SIMD, MAX_CLOCK_GHZ, ISSUE_PER_CLK_PER_SIMD = 1024, 2.4, 0.5
LANE_INSTR_PER_S = N_SIMD * MAX_CLOCK_GHZ * 1e9 * ISSUE_PER_CLK_PER_SIMD * 64
# minimal vector instructions per operation: Blake2s = 10 rounds x 8 G x 12 (4 three-operand adds, 4 xors, 4 rotates)
# + 16 for the feed-forward; M31 butterfly = 11 with doubled twiddles (fft_fixed.hip)
INSTR_PER_COMPRESSION, INSTR_PER_BUTTERFLY = 12 * 80 + 16, 11

KERNEL_FAMILIES = {
    # name: (timings prefix, rocprof kernel names, operation count field, measured ceiling in G ops/s, unit, instr/op)
    "k_fft_fx": ("fft", ["k_fft_staged<false>", "k_fft_staged<true>", "k_fft_interp_extend", "k_fft_fx", "k_fft_interp_extend_fx"],
                 "fft_butterflies", BUTTERFLY_PEAK_G, "G butterflies/s", INSTR_PER_BUTTERFLY),
    "k_merkle_fused": ("merkle_fused", ["k_merkle_fused", "k_merkle_fused<0>", "k_merkle_fused<1>", "k_merkle_fused<2>",
                                        "k_merkle_fused<3>", "k_merkle_fused<4>"],
                       "merkle_fused_compressions", BLAKE2S_PEAK_GCOMP, "G Blake2s compressions/s", INSTR_PER_COMPRESSION),
}


PMC_ROUND = "r6"   # the round whose tree the committed counter summary must describe


def load_pmc(root: str):
    """(per-kernel PMC summary, its description) - the rocprofv3 counter summary committed for THIS round's tree
    (profiles/<PMC_ROUND>_pmc_summary.json).  If it is absent the newest older summary is still used - the counters of the
    Merkle and transform kernels move little between rounds - but never silently: the description (which becomes the
    line's `traffic_source`) names the round the figures come from and says STALE, and a warning goes to stderr."""
    import glob
    import re
    import sys
    cur = os.path.join(root, "profiles", "%s_pmc_summary.json" % PMC_ROUND)
    older = sorted((p_ for p_ in glob.glob(os.path.join(root, "profiles", "r*_pmc_summary.json")) if p_ != cur),
                   key=lambda p_: int(re.search(r"r(\d+)_pmc_summary", p_).group(1)), reverse=True)
    for pth in [cur] + older:
        if not os.path.exists(pth):
            continue
        try:
            with open(pth) as f:
                kernels = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        name = "profiles/" + os.path.basename(pth)
        if pth != cur:
            sys.stderr.write("bench.py: WARNING: %s is missing; `traffic` comes from the STALE counter summary %s\n"
                             % (os.path.relpath(cur, root), name))
            name += " [STALE: counters of an earlier round's tree, this round's summary (%s) is missing]" % os.path.basename(cur)
        return kernels, name
    return {}, None


def kernel_roofline(name: str, tm: Dict[str, float], pmc: dict, pmc_source: Optional[str]) -> dict:
    """`roofline` object of one kernel family from the library's own per-launch accounting of a profiled solo proof
    (`lmn_timings`: HIP events on the prover's stream around every launch of the family)."""
    prefix, pmc_names, ops_field, alu_peak, alu_unit, instr_per_op = KERNEL_FAMILIES[name]
    ms, nbytes, launches = tm[prefix + "_ms"], tm[prefix + "_bytes"], max(int(tm[prefix + "_launches"]), 1)
    sec = 1e-3 * ms
    achieved = nbytes / sec / 1e9 if ms > 0 else 0.0
    traffic = None
    got = [v for k, v in pmc.items() if any(k == n or k.startswith(n + "<") for n in pmc_names)]
    if got:
        tot_l = sum(g["launches"] for g in got)
        traffic = sum(g["hbm_bytes_per_launch_corrected"] * g["launches"] for g in got) / max(tot_l, 1)
    ops = tm[ops_field]
    alu_achieved = ops / sec / 1e9 if ms > 0 else 0.0
    arch_peak = LANE_INSTR_PER_S / instr_per_op / 1e9
    counter_gbs = traffic * launches / sec / 1e9 if (traffic and ms > 0) else None
    # what the kernel is bound by: the ceiling it sits closest to
    valu_bound = alu_achieved / alu_peak > achieved / HBM_PEAK_GBS
    return {
        "bound": "valu" if valu_bound else "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "traffic_vs_algorithmic": (traffic / (nbytes / launches)) if traffic and nbytes else None,
        "achieved_by_counter_traffic": counter_gbs,
        "frac_by_counter_traffic": (counter_gbs / HBM_PEAK_GBS) if counter_gbs is not None else None,
        "frac_note": "`frac` prices SURVEY 8(d)'s algorithmic bytes (every tree node written and re-read); "
                     "`frac_by_counter_traffic` the bytes the PMC passes saw this kernel move; `alu_ceiling` is what binds it"
        if valu_bound else None,
        "traffic_source": (pmc_source + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH+WRITE)")
        if traffic is not None and pmc_source else None,
        "launches_per_proof": launches, "avg_launch_ms": ms / launches, "algorithmic_bytes_per_launch": nbytes / launches,
        "alu_ceiling": {"achieved": alu_achieved, "peak_measured": alu_peak, "unit": alu_unit, "frac": alu_achieved / alu_peak,
                        "architectural": {"vector_instructions_per_op_min": instr_per_op,
                                          "issue": "%.1f wave-instr/clk/SIMD x %d SIMDs x %.1f GHz x 64 lanes"
                                                   % (ISSUE_PER_CLK_PER_SIMD, N_SIMD, MAX_CLOCK_GHZ),
                                          "peak": arch_peak, "frac": alu_achieved / arch_peak}},
    }


def whole_proof(log_rows: int, ms_per_proof: float, world: int, counter_bytes_per_proof: Optional[float] = None) -> dict:
    """Whole proof against SURVEY.md §8(d)'s minimum-traffic model (48*C*N + 1500*N bytes, C = 27 columns: every pass of
    every stage counted once - not the per-pass Merkle formula `roofline` uses for its launches)."""
    model_bytes = (48 * 27 + 1500) * float(1 << log_rows)
    out = {"byte_model": "SURVEY.md §8(d) whole-proof minimum-traffic model: 48*C*N + 1500*N bytes, C = 27, N = 2^%d "
                         "(all stages; differs from the per-launch Merkle bytes behind `roofline`)" % log_rows,
           "model_bytes_per_proof": model_bytes, "achieved": model_bytes / (1e-3 * ms_per_proof) / 1e9 * world,
           "peak": HBM_PEAK_GBS * world, "unit": "GB/s"}
    out["frac"] = out["achieved"] / out["peak"]
    if counter_bytes_per_proof:
        out["counter_bytes_per_proof"] = counter_bytes_per_proof
        out["frac_by_counter_traffic"] = counter_bytes_per_proof / (1e-3 * ms_per_proof) / 1e9 * world / out["peak"]
    return out
