"""VALU issue cost per instruction class on THIS box (wsnark_peak_probe 6..15): the lane-operation rate of every class of
tools/isa_histogram.py, the shader clock under the multiply-add load, and each class as cycles per wave-instruction per SIMD.
Usage: python tools/issue_probe.py [reps] > gpurun_out/<TAG>/issue_classes.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PROBES = (("bit32", 6, "v_add_u32"), ("shift64", 7, "v_lshrrev_b64"), ("bit32_and", 8, "v_and_b32"), ("mul32", 9, "v_mul_lo_u32"),
          ("mad64", 10, "v_mad_u64_u32"), ("add64c", 11, "v_add_co_u32 + v_addc_co_u32 (per instruction)"), ("mov", 12, "v_mov_b32"),
          ("select", 13, "v_cndmask_b32 on a standing vcc"), ("dpp", 14, "v_mov_b32_dpp quad_perm"), ("compare", 16, "v_cmp_lt_u32 into SGPR pairs"),
          ("select_then_add", 18, "v_cndmask_b32 (standing vcc) alternating with independent v_add_u32 (per instruction)"),
          ("select_sgpr", 19, "v_cndmask_b32_e64 with the mask in an SGPR pair"), ("select_bitwise", 20, "xor/and/xor x2 + v_bfi_b32 x2 (per instruction)"),
          ("vop3_3src", 21, "v_add3_u32"), ("bfi", 22, "v_bfi_b32"), ("select_pairs", 23, "v_cndmask_b32 in runs of two between v_add_u32 pairs (per instruction)"),
          ("alignbit", 24, "v_alignbit_b32"),
          ("and_literal", 25, "v_and_b32 with a 32-bit literal (8-byte encoding)"), ("add_e64", 26, "v_add_u32 forced into the VOP3 encoding"),
          ("and_sgpr", 27, "v_and_b32 with an SGPR source (4-byte encoding)"), ("shift32_inline", 28, "v_lshrrev_b32 by an inline constant"),
          ("cmp_sel", 17, "v_cmp_lt_u32 vcc + the v_cndmask_b32 that reads it, back to back (per instruction)"))


def measure(bn, reps=3):
    """{class: G lane-ops/s (best of reps)}, clock_GHz, and cycles per wave-instruction per SIMD"""
    def probe(p):
        best = 0.0
        for _ in range(reps):
            v = C.c_double(0)
            bn.lib.check(bn.lib.c.wsnark_peak_probe(p, C.byref(v)))
            best = max(best, v.value)
        return best
    rates = {name: round(probe(p), 1) for name, p, _ in PROBES}
    try:
        import torch
        n_simd = torch.cuda.get_device_properties(0).multi_processor_count * 4
    except Exception:  # noqa: BLE001
        n_simd = 1024
    # cycles: a plain 32-bit VALU instruction issues a wave64 over 2 cycles on gfx950's SIMD-32 (MI355X_MICROARCH.md "Wave scheduling");
    # the clock that makes the measured v_add_u32 rate exactly that is the clock the other classes are expressed in
    clock = rates["bit32"] * 1e9 * 2 / (64 * n_simd) / 1e9 if rates.get("bit32") else 0.0
    out = {"G_lane_ops_per_s": rates, "instruction": {name: ins for name, _, ins in PROBES}, "n_simd": n_simd,
           "clock_GHz_implied_by_2_cycle_v_add_u32": round(clock, 3),
           "modmul_inlined_G_per_s": round(probe(1), 1), "mad_u64_u32_c_loop_G_per_s": round(probe(2), 1)}
    if clock > 0:
        out["cycles_per_wave_instruction"] = {k: round(clock * 1e9 * n_simd * 64 / (v * 1e9), 2) if v else None for k, v in rates.items()}
    out["relative_to_mad64"] = {k: round(rates["mad64"] / v, 3) if v else None for k, v in rates.items()}
    return out


if __name__ == "__main__":
    import wasmsnark_amd
    bn = wasmsnark_amd.build(device=0)
    print(json.dumps(measure(bn, int(sys.argv[1]) if len(sys.argv) > 1 else 3)))
