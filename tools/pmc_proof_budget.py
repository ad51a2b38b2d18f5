#!/usr/bin/env python3
"""Per-proof instruction budget from a rocprofv3 --pmc pass around tools/proof_counters.py.
  tools/pmc_proof_budget.py <dir with *counter_collection.csv> <proofs between the markers> [n_simd] [n_xcd]
Only the dispatches BETWEEN the two groups of marker launches (probe_inverse_kernel) are counted.  Output (JSON):
  kernels[name] = {launches_per_proof, valu_insts_per_proof, waves_per_proof, busy_cycles_per_proof (GRBM_GUI_ACTIVE / n_xcd, summed
                   over the kernel's launches: what the launches would take one after the other), issue_floor_ms_per_proof, valu_slots_flat4}
  valu_insts_per_proof   sum over kernels of SQ_INSTS_VALU (wave-instructions)
  issue_floor_ms_per_proof   (round 6) sum over kernels of their issue floors BY INSTRUCTION CLASS (tools/issue_model.py: static class
                         mix x class rates measured on the box, --rates <issue_classes.json>): what a chip that did nothing but issue
                         this proof's instruction stream would need, whatever the schedule.
  issue_cycles_flat4_per_proof   the round-3..5 figure: 4 x valu_insts_per_proof / n_simd (every instruction priced at 4 cycles: too high)
bench.py re-prices the per-kernel counts with the rates of ITS run and divides by one measured proof: `roofline_proof.frac`.
  tools/pmc_proof_budget.py <dir> <proofs> [n_simd [n_xcd]] [--rates issue_classes.json] [--classes isa_classes.json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import issue_model  # noqa: E402

argv = list(sys.argv[1:])
opt = {}
for flag in ("--rates", "--classes"):
    if flag in argv:
        i = argv.index(flag)
        opt[flag] = argv[i + 1]
        del argv[i:i + 2]
root = argv[0]
P = int(argv[1])
n_simd = int(argv[2]) if len(argv) > 2 else 1024
n_xcd = int(argv[3]) if len(argv) > 3 else 8
classes = issue_model.load_classes(opt.get("--classes"))
rates, rates_src = dict(issue_model.DEFAULT_RATES), "tools/issue_model.py DEFAULT_RATES (gpurun call r06_c05)"
if opt.get("--rates") and os.path.exists(opt["--rates"]):
    try:
        rates.update(json.load(open(opt["--rates"]))["G_lane_ops_per_s"])
        rates_src = opt["--rates"]
    except Exception:  # noqa: BLE001
        pass
rows = []
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        rows += list(csv.DictReader(fh))
by_dispatch = defaultdict(dict)
names = {}
for r in rows:
    d = int(r["Dispatch_Id"])
    names[d] = r["Kernel_Name"]
    try:
        by_dispatch[d][r["Counter_Name"]] = float(r["Counter_Value"])
    except ValueError:
        pass
marks = sorted(d for d, n in names.items() if "probe_inverse_kernel" in n)
if len(marks) < 2:
    sys.exit("no marker launches found")
gap = max(range(1, len(marks)), key=lambda i: marks[i] - marks[i - 1])      # the two groups are separated by the proofs
lo, hi = marks[gap - 1], marks[gap]


def short(name):
    m = re.search(r"wsnark::([A-Za-z0-9_]+)", name)
    s = m.group(1) if m else name.split("(")[0].replace("void ", "").strip()[:60]
    if m and ("Fp2" in name or "Fe2T" in name):
        s += "_g2"
    return s


acc = defaultdict(lambda: defaultdict(float))
for d, ctr in by_dispatch.items():
    if not (lo < d < hi):
        continue
    k = short(names[d])
    acc[k]["launches"] += 1
    for c, v in ctr.items():
        acc[k][c] += v
out, total_insts, total_busy, total_floor = {}, 0.0, 0.0, 0.0
for k, a in sorted(acc.items()):
    insts, busy = a.get("SQ_INSTS_VALU", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0) / n_xcd
    total_insts += insts
    total_busy += busy
    dyn64 = a["SQ_INSTS_VALU_INT64"] / insts if insts and "SQ_INSTS_VALU_INT64" in a else None
    floor_s, how = issue_model.kernel_floor_s(k, insts / P, classes, rates, dyn64)
    total_floor += floor_s
    out[k] = {"launches_per_proof": round(a["launches"] / P, 2), "valu_insts_per_proof": round(insts / P, 1),
              "waves_per_proof": round(a.get("SQ_WAVES", 0.0) / P, 1), "busy_cycles_per_proof": round(busy / P, 1),
              "issue_floor_ms_per_proof": round(floor_s * 1e3, 4), "issue_floor_how": how,
              # DYNAMIC check of the static class mix (when the pass collected them): the hardware's own split of the VALU instructions
              # into 32-bit and 64-bit integer operations, beside the static mix's share of multiply-adds + 64-bit shifts / moves
              "dynamic_int64_share": round(a["SQ_INSTS_VALU_INT64"] / insts, 4) if insts and "SQ_INSTS_VALU_INT64" in a else None,
              "dynamic_int32_share": round(a["SQ_INSTS_VALU_INT32"] / insts, 4) if insts and "SQ_INSTS_VALU_INT32" in a else None,
              "static_mad64_plus_wide64_share": (round(classes["kernels"][k]["mix"]["mad64"] + classes["kernels"][k]["mix"]["wide64"], 4)
                                                 if classes and k in classes["kernels"] else None),
              "valu_slots_flat4": round(4 * insts / (busy * n_simd), 4) if busy else None}
print(json.dumps({"how": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace around tools/proof_counters.py; dispatches between "
                         "the two marker groups only; kernels are serialised by the counter collection, so busy_cycles are each kernel ALONE",
                  "proofs": P, "n_simd": n_simd, "n_xcd": n_xcd, "dispatches_counted": sum(int(a["launches"]) for a in acc.values()),
                  "valu_insts_per_proof": round(total_insts / P, 1), "issue_floor_ms_per_proof": round(total_floor * 1e3, 4),
                  "class_rates_G_lane_ops_per_s": rates, "class_rates_source": rates_src, "class_mix_source": os.path.basename(classes["_path"]) if classes else None,
                  "issue_cycles_flat4_per_proof": round(4 * total_insts / P / n_simd, 1),
                  "serialised_busy_cycles_per_proof": round(total_busy / P, 1), "kernels": out}, indent=1))
