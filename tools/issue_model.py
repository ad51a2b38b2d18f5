"""The VALU issue roofline by INSTRUCTION CLASS (round 6; replaces the flat "4 cycles per VALU instruction" of rounds 3-5, under which
mul_base_kernel read 1.34 of the chip's issue slots).

A kernel's floor is the time a chip that did nothing but issue its instruction stream would need:

    floor_s = SQ_INSTS_VALU x 64 lanes x sum over classes ( share_of_class / rate_of_class )

  * SQ_INSTS_VALU: the kernel's DYNAMIC wave-instruction count (rocprofv3 --pmc);
  * share_of_class: the kernel's STATIC mix by issue class (tools/isa_histogram.py -> profiles/rNN_isa_classes.json; operand-aware:
    an SGPR source or a third VGPR source makes a plain 32-bit instruction cost what a multiply-add costs);
  * rate_of_class: lane-operations per second of that class MEASURED on the box (wsnark_peak_probe 6.., tools/issue_probe.py ->
    `issue_classes` in the bench line): eight independent chains per lane, eight wavefronts per SIMD, inline assembly.
No clock enters: rates are wall-clock rates.  MI355X, round 6: ~70 T lane-ops/s for plain one- and two-source 32-bit instructions
(2 cycles per wave64 on the SIMD-32), ~37.5 T/s (x 1.87) for v_mad_u64_u32, v_mul_lo_u32, 64-bit shifts, carry and compare
instructions, DPP, three-source VOP3 and anything with an SGPR source."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# fallback rates (G lane-ops/s), MI355X, gpurun call r06_c05 -- used only when a run has no probe figures of its own
DEFAULT_RATES = {"bit32": 69600.0, "shift64": 37860.0, "mul32": 37730.0, "mad64": 37250.0, "add64c": 37650.0, "mov": 70040.0, "select": 6870.0,
                 "dpp": 37870.0, "compare": 37980.0, "and_sgpr": 38100.0, "vop3_3src": 37400.0}


def latest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return fs[-1] if fs else None


def load_classes(path=None):
    path = path or latest("r*_isa_classes.json")
    if not path:
        return None
    d = json.load(open(path))
    d["_path"] = path
    return d


def seconds_per_wave_instruction(mix, probe_of, rates):
    """mean time one wave-instruction of this mix occupies the CHIP (all SIMDs issuing): sum share / (rate / 64)"""
    t = 0.0
    for cls, share in mix.items():
        r = rates.get(probe_of.get(cls, "bit32")) or DEFAULT_RATES.get(probe_of.get(cls, "bit32")) or DEFAULT_RATES["bit32"]
        t += share * 64.0 / (r * 1e9)
    return t


def corrected_mix(mix, dyn64):
    """The static mix with its 64-bit share (multiply-adds + 64-bit shifts / moves) replaced by the share the HARDWARE counted
    (SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU of the same dispatches): the two agree within 2 % for the straight-line accumulation loops and
    differ where a static count cannot know how often a call site or a branch runs (the transform passes' product calls: 0.69 static,
    0.49 counted).  The other classes keep their proportions among themselves."""
    s64 = mix.get("mad64", 0.0) + mix.get("wide64", 0.0)
    if dyn64 is None or s64 <= 0 or s64 >= 1:
        return mix
    out = {}
    for c, v in mix.items():
        out[c] = v * dyn64 / s64 if c in ("mad64", "wide64") else v * (1.0 - dyn64) / (1.0 - s64)
    return out


def kernel_floor_s(kernel, valu_insts, classes, rates, dyn64=None):
    """issue floor (seconds) of `valu_insts` wave-instructions of kernel `kernel`; (floor, how) -- unknown kernels are priced at the
    multiply-add rate (the conservative end: a floor that is too high makes a fraction too high, never hides a gap... it is flagged).
    dyn64: the dispatches' measured share of 64-bit integer instructions, when the counter pass collected it (corrected_mix)."""
    k = classes["kernels"].get(kernel) if classes else None
    if k is None:
        r = rates.get("mad64") or DEFAULT_RATES["mad64"]
        return valu_insts * 64.0 / (r * 1e9), "unclassified: priced at the multiply-add rate"
    mix = corrected_mix(k["mix"], dyn64)
    how = "class mix of %s" % k["function"] + ("" if dyn64 is None else ", 64-bit share as counted (%.3f; static %.3f)" % (dyn64, k["mix"].get("mad64", 0) + k["mix"].get("wide64", 0)))
    return valu_insts * seconds_per_wave_instruction(mix, classes["probe_of_class"], rates), how


def equivalent_cycles(rates, n_simd=1024):
    """each class as cycles per wave-instruction and SIMD, in the clock that makes v_add_u32 exactly 2 cycles (informational)"""
    b = rates.get("bit32") or DEFAULT_RATES["bit32"]
    clock_hz = b * 1e9 * 2 / (64 * n_simd)
    return {k: round(clock_hz * n_simd * 64 / (v * 1e9), 2) for k, v in rates.items() if v}, round(clock_hz / 1e9, 3)
