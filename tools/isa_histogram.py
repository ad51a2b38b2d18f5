"""Per-kernel VALU opcode histogram of the shipped device code, by ISSUE CLASS (no GPU needed).

hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S on every .hip source with the flags of wasmsnark_amd/csrc/Makefile, then
the instructions between a kernel's label and its s_endpgm are counted.  The counts are STATIC (an instruction inside a loop counts
once); the hot kernels of this library are straight-line bodies inside one loop, so the static mix of a kernel is the mix of what
it issues -- bench.py / tools/pmc_proof_budget.py multiply that mix by the DYNAMIC total (SQ_INSTS_VALU) and by each class's
measured cost (wsnark_peak_probe 6.., `issue_classes` in the bench line).

Usage: python tools/isa_histogram.py [--json] > profiles/rNN_isa_classes.{md,json}"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wasmsnark_amd", "csrc")
SRCS = ["ntt.hip", "msm.hip", "calch.hip", "dist.hip", "fixedbase.hip", "selftest.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S"]

# issue classes: each has a probe in selftest.hip (wsnark_peak_probe) that measures its wave-instruction rate on the box
CLASSES = ["mad64", "mul32", "shift64", "add64c", "bit32", "mov", "cmp_sel", "dpp", "other"]


def classify(op):
    """VALU mnemonic -> issue class (the probe that prices it)."""
    base = op.split("_e32")[0].split("_e64")[0]
    if base.endswith("_dpp") or "_dpp" in op or base.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_bpermute")):
        return "dpp"
    if base.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "mad64"
    if base.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mad_u32_u16", "v_mul_hi_u32_u24")):
        return "mul32"
    if base.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_add_u64")):
        return "shift64"
    if base.startswith(("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32")):
        return "add64c"
    if base.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr", "v_swap")):
        return "mov"
    if base.startswith(("v_cmp", "v_cndmask", "v_cmpx")):
        return "cmp_sel"
    if base.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_lshlrev_b32",
                        "v_ashrrev_i32", "v_and_or_b32", "v_lshl_or_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_add3_u32", "v_or3_b32", "v_alignbit_b32",
                        "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_xad_u32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_bfm_b32", "v_alignbyte",
                        "v_perm_b32", "v_mbcnt", "v_ffbh", "v_ffbl", "v_bcnt", "v_sub_i32", "v_add_i32", "v_cvt", "v_lshl_add", "v_add_nc", "v_xnor")):
        return "bit32"
    return "other"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("wsnark::", "").replace("void ", "")
    for a, b in (("Curve<Fp2T<Field29<Fq29Params> > >", "G2R29"), ("Curve<Field29I<Fq29Params> >", "G1R29I"), ("Curve<Field29<Fq29Params> >", "G1R29"),
                 ("Curve<Fp2PairT<Field29<Fq29Params> > >", "G2P29"), ("Curve<Fp2T<Field<FqParams> > >", "G2"), ("Curve<Field<FqParams> >", "G1"),
                 ("Field29<Fr29Params>", "Fr29"), ("Field<FrParams>", "Fr")):
        name = name.replace(a, b)
    return name


def one(src):
    out = "/tmp/isa_hist_%s.s" % src.replace(".", "_")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], cwd=CSRC, check=True, capture_output=True)
    txt = open(out).read()
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", txt, re.M))
    rows = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.split() and not l.strip().startswith((".", ";"))]
        h = dict.fromkeys(CLASSES, 0)
        ops = {}
        n_valu = 0
        for i in ins:
            if not i.startswith("v_"):
                continue
            n_valu += 1
            c = classify(i)
            h[c] += 1
            if c == "other":
                ops[i] = ops.get(i, 0) + 1
        rows.append({"src": src, "name": name, "kernel": name in kernels, "valu": n_valu, "all": len(ins),
                     "salu": sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_waitcnt", "s_nop"))),
                     "s_nop": sum(1 for i in ins if i.startswith("s_nop")),
                     "mem": sum(1 for i in ins if i.startswith(("global_", "buffer_", "flat_", "scratch_"))),
                     "lds": sum(1 for i in ins if i.startswith("ds_")), "classes": h, "other_ops": ops})
    return rows


def histogram():
    with ThreadPoolExecutor(len(SRCS)) as ex:
        rows = [r for rs in ex.map(one, SRCS) for r in rs]
    dm = demangle([r["name"] for r in rows])
    for r in rows:
        r["short"] = short(dm.get(r["name"], r["name"]))
    return rows


def main():
    rows = histogram()
    if "--json" in sys.argv:
        print(json.dumps({r["short"]: {k: r[k] for k in ("src", "kernel", "valu", "all", "salu", "s_nop", "mem", "lds", "classes", "other_ops")} for r in rows}, indent=1))
        return
    print("Static VALU opcode mix per device function, by issue class (tools/isa_histogram.py; functions that are not kernels are the")
    print("non-inlined product bodies the kernels call).  `other` lists its opcodes.\n")
    print("| source | function | kernel | VALU | " + " | ".join(CLASSES) + " | SALU | s_nop | mem | LDS | other opcodes |")
    print("|---|---|---|---|" + "---|" * (len(CLASSES) + 5))
    for r in rows:
        print("| %s | `%s` | %s | %d | " % (r["src"], r["short"], "yes" if r["kernel"] else "", r["valu"]) + " | ".join(str(r["classes"][c]) for c in CLASSES)
              + " | %d | %d | %d | %d | %s |" % (r["salu"], r["s_nop"], r["mem"], r["lds"], " ".join("%s:%d" % kv for kv in sorted(r["other_ops"].items()))))


if __name__ == "__main__":
    sys.exit(main())
