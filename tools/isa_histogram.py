"""Per-kernel VALU instruction mix of the shipped device code, by ISSUE CLASS (no GPU needed).

hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S on every .hip source with the flags of wasmsnark_amd/csrc/Makefile; the
instructions between a function's label and its end are classified by what the issue-class probes (wsnark_peak_probe 6.., tools/
issue_probe.py) showed to decide a wave64 VALU instruction's cost on gfx950 -- not its mnemonic alone but its OPERANDS:

  mad64      v_mad_u64_u32 / v_mad_i64_i32                                   (~3.7 cycles per wave-instruction)
  mul32      v_mul_lo / v_mul_hi                                             (~3.7)
  wide64     64-bit shifts, adds and moves                                   (~3.7)
  carry      v_add_co / v_addc_co / v_sub_co / ...  (carry in or out of an SGPR pair)   (~3.7 each)
  compare    v_cmp* (writes vcc or an SGPR pair)                             (~3.7)
  sgpr_src   any other VALU instruction with an SGPR source operand          (~3.7: v_and_b32 v, s, v is TWICE v_and_b32 v, v, v)
  src3       any other VALU instruction with three VGPR sources (v_add3_u32, v_or3_b32, v_bfi_b32, v_alignbit_b32 ...)   (~3.7)
  dpp        DPP / SDWA / lane-crossing forms                                (~3.7)
  fast       everything else: one or two VGPR sources, inline constants or a 32-bit literal, VOP2 or VOP3 encoding alike   (2.0)
  select_run v_cndmask_b32_e32 in runs of three or more (the members beyond the second): ~20 cycles each back to back on a standing
             vcc, 2 when other instructions sit between them -- counted apart so that a kernel with long runs shows it

The counts are STATIC, weighted by LOOP DEPTH: the compiler annotates every basic block with the depth of the loop it belongs to, and an
instruction at depth d counts 16^d times -- so the mix of a kernel with a loop is the mix of its loop body (the accumulation's
per-addition code, not its prologue; the fixed-base helper's double-and-add, not its setup).  A kernel's non-inlined callees (the
Montgomery products of the transform and tail kernels: `s_swappc_b64`) are added per call site with the site's weight.  The columns
of the table are the unweighted counts; `mix` in the JSON is the weighted share.  bench.py / tools/pmc_counters.py /
tools/pmc_proof_budget.py multiply the mix by the DYNAMIC total (SQ_INSTS_VALU) and divide by each class's measured rate.

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

CLASSES = ["mad64", "mul32", "wide64", "carry", "compare", "sgpr_src", "src3", "dpp", "fast", "select_run"]
# class -> the probe of tools/issue_probe.py whose rate prices it
PROBE_OF = {"mad64": "mad64", "mul32": "mul32", "wide64": "shift64", "carry": "add64c", "compare": "compare", "sgpr_src": "and_sgpr",
            "src3": "vop3_3src", "dpp": "dpp", "fast": "bit32", "select_run": "select"}

_WIDE = ("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_add_u64", "v_mov_b64", "v_and_b64", "v_or_b64", "v_xor_b64")
_CARRY = ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32")
_MUL = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u16", "v_mul_f", "v_fma", "v_rcp", "v_trunc", "v_fmamk", "v_cvt")
_LANE = ("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_bpermute", "ds_bpermute", "ds_swizzle")


def classify(line):
    """one assembly line (mnemonic + operands) of a VALU instruction -> issue class"""
    parts = line.strip().split(None, 1)
    op = parts[0]
    ops = (parts[1] if len(parts) > 1 else "").split(";")[0]
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "mad64"
    if op.startswith(_MUL):
        return "mul32"
    if op.startswith(_WIDE):
        return "wide64"
    if op.startswith(_CARRY):
        return "carry"
    if op.startswith("v_cmp"):
        return "compare"
    if "_dpp" in op or "_sdwa" in op or "quad_perm" in ops or "row_" in ops or op.startswith(_LANE):
        return "dpp"
    toks = [t.strip() for t in ops.split(",")]
    srcs = toks[1:]
    n_v = sum(1 for t in srcs if re.match(r"^-?\|?v(\d+|\[)", t))
    n_s = sum(1 for t in srcs if re.match(r"^s(\d+|\[)", t) or t in ("vcc_lo", "vcc_hi", "m0", "exec_lo", "exec_hi"))
    if op.startswith("v_cndmask_b32_e64") or n_s:
        return "sgpr_src"
    if n_v >= 3:
        return "src3"
    return "fast"


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


def pmc_name(demangled):
    """the name tools/pmc_counters.py / pmc_proof_budget.py give the kernel's dispatches"""
    m = re.search(r"wsnark::([A-Za-z0-9_]+)", demangled)
    s = m.group(1) if m else demangled.split("(")[0].replace("void ", "").strip()[:60]
    head = demangled.split("(")[0]
    if m and ("Fp2" in demangled or "Fe2T" in demangled):
        s += "_g2"
    del head
    return s


def one(src):
    out = "/tmp/isa_hist_%s.s" % src.replace(".", "_")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], cwd=CSRC, check=True, capture_output=True)
    txt = open(out).read()
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", txt, re.M))
    rows = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [l for l in body.splitlines() if l.startswith("\t") and l.split() and not l.strip().startswith((".", ";"))]
        h = dict.fromkeys(CLASSES, 0)
        hw = dict.fromkeys(CLASSES, 0.0)          # the same, every instruction weighted 16^(loop depth of its block)
        run, depth, calls_w = 0, 0, 0.0
        for raw in body.splitlines():
            mlab = re.match(r"^\.LBB\w+:\s*(;.*)?$", raw)
            if mlab:                                # a block label: "; =>This Inner Loop Header: Depth=1" / ";   in Loop: Header=BB3_6 Depth=2" / none
                md = re.findall(r"Depth=(\d+)", raw)
                depth = max(int(x) for x in md) if md else 0
                continue
            if not raw.startswith("\t") or not raw.split() or raw.strip().startswith((".", ";")):
                continue
            l, w = raw, 16.0 ** depth
            op = l.split()[0]
            if op.startswith("s_swappc"):
                calls_w += w
            if not op.startswith("v_"):
                if not op.startswith(("s_nop", "s_waitcnt")):
                    run = 0 if not op.startswith("s_") else run      # (scalar instructions issue from another port: a run survives them)
                continue
            if op.startswith("v_cndmask_b32_e32"):
                run += 1
                c = "select_run" if run >= 3 else "fast"
            else:
                run = 0
                c = classify(l)
            h[c] += 1
            hw[c] += w
        ops = [l.split()[0] for l in lines]
        callees = re.findall(r"(_ZN6wsnark\w+)@rel32@lo", body)
        rows[name] = {"src": src, "name": name, "kernel": name in kernels, "classes": h, "weighted": hw, "calls_weight": calls_w,
                      "valu": sum(h.values()), "all": len(ops),
                      "calls": sum(1 for o in ops if o.startswith("s_swappc")), "callees": callees,
                      "salu": sum(1 for o in ops if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop"))),
                      "s_nop": sum(1 for o in ops if o.startswith("s_nop")),
                      "mem": sum(1 for o in ops if o.startswith(("global_", "buffer_", "flat_", "scratch_"))),
                      "lds": sum(1 for o in ops if o.startswith("ds_"))}
    # a kernel's calls: dealt to the callees it references, in proportion to the references (one reference may serve several sites)
    for r in rows.values():
        r["with_callees"] = dict(r["weighted"])
        refs = [c for c in r["callees"] if c in rows and not rows[c]["kernel"]]
        if r["calls"] and refs:
            for c in set(refs):
                share = r["calls_weight"] * refs.count(c) / len(refs)
                for k, v in rows[c]["classes"].items():      # (the callees are straight-line products: unweighted counts)
                    r["with_callees"][k] += v * share
        r["valu_with_callees"] = sum(r["with_callees"].values())
    return list(rows.values())


def histogram():
    with ThreadPoolExecutor(len(SRCS)) as ex:
        rows = [r for rs in ex.map(one, SRCS) for r in rs]
    dm = demangle([r["name"] for r in rows])
    for r in rows:
        r["demangled"] = dm.get(r["name"], r["name"])
        r["short"] = short(r["demangled"])
        r["pmc_name"] = pmc_name(r["demangled"])
    return rows


def as_json(rows):
    """{pmc kernel name: {"mix": {class: share of the VALU stream}, ...}}: the first instantiation seen stands for a name"""
    out = {}
    # several instantiations share a dispatch name (msm_tree on one lane / on lane pairs; 4- and 8-byte grouping entries): the one the
    # default build LAUNCHES goes first
    rows = sorted(rows, key=lambda r: 0 if "CurvePairG1" in r["demangled"] and ("msm_tree" in r["demangled"] or "msm_rows" in r["demangled"]) else 1)
    for r in rows:
        if not r["kernel"] or not r["valu_with_callees"]:
            continue
        tot = float(r["valu_with_callees"])
        out.setdefault(r["pmc_name"], {"function": r["short"], "src": r["src"], "static_valu": r["valu"], "static_valu_with_callees": round(tot, 1),
                                       "call_sites": r["calls"], "mix": {k: round(v / tot, 5) for k, v in r["with_callees"].items()},
                                       "salu": r["salu"], "s_nop": r["s_nop"], "mem": r["mem"], "lds": r["lds"]})
    return {"how": "tools/isa_histogram.py: static VALU mix per kernel by issue class, instructions weighted 16^(loop depth), non-inlined callees added per call site with the site's weight",
            "classes": CLASSES, "probe_of_class": PROBE_OF, "kernels": out}


def main():
    rows = histogram()
    if "--json" in sys.argv:
        print(json.dumps(as_json(rows), indent=1))
        return
    print("Static VALU instruction mix per device function, by issue class (tools/isa_histogram.py -- the classes and what decides them")
    print("are in that file's header).  `calls` = non-inlined product calls in the body.  `slow share` = the share of the LOOP-WEIGHTED stream (callees")
    print("included) that is not in the 2-cycle class -- what the issue model prices.\n")
    print("| source | function | kernel | VALU | " + " | ".join(CLASSES) + " | calls | weighted stream (relative) | slow share | SALU | s_nop | mem | LDS |")
    print("|---|---|---|---|" + "---|" * (len(CLASSES) + 7))
    for r in rows:
        tot = r["valu_with_callees"] or 1
        slow = 1.0 - (r["with_callees"]["fast"]) / tot
        print("| %s | `%s` | %s | %d | " % (r["src"], r["short"], "yes" if r["kernel"] else "", r["valu"]) + " | ".join(str(r["classes"][c]) for c in CLASSES)
              + " | %d | %d | %.3f | %d | %d | %d | %d |" % (r["calls"], round(r["valu_with_callees"]), slow, r["salu"], r["s_nop"], r["mem"], r["lds"]))


if __name__ == "__main__":
    sys.exit(main())
