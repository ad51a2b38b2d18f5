#!/usr/bin/env python3
"""Per-proof instruction budget from a rocprofv3 --pmc pass around tools/proof_counters.py.
  tools/pmc_proof_budget.py <dir with *counter_collection.csv> <proofs between the markers> [n_simd] [n_xcd]
Only the dispatches BETWEEN the two groups of marker launches (probe_inverse_kernel) are counted.  Output (JSON):
  kernels[name] = {launches_per_proof, valu_insts_per_proof, waves_per_proof, busy_cycles_per_proof (GRBM_GUI_ACTIVE / n_xcd, summed
                   over the kernel's launches: what the launches would take one after the other), valu_issue_frac (of that)}
  valu_insts_per_proof   sum over kernels of SQ_INSTS_VALU (wave-instructions)
  issue_cycles_per_proof 4 x valu_insts_per_proof / n_simd: cycles of a chip that issues one wave64 VALU instruction per SIMD16 every
                         4 cycles and does nothing else -- the floor of this instruction stream, whatever the schedule.
bench.py divides that floor by the cycles of one measured proof: `roofline_proof.frac`."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
P = int(sys.argv[2])
n_simd = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
n_xcd = int(sys.argv[4]) if len(sys.argv) > 4 else 8
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
out, total_insts, total_busy = {}, 0.0, 0.0
for k, a in sorted(acc.items()):
    insts, busy = a.get("SQ_INSTS_VALU", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0) / n_xcd
    total_insts += insts
    total_busy += busy
    out[k] = {"launches_per_proof": round(a["launches"] / P, 2), "valu_insts_per_proof": round(insts / P, 1),
              "waves_per_proof": round(a.get("SQ_WAVES", 0.0) / P, 1), "busy_cycles_per_proof": round(busy / P, 1),
              "valu_issue_frac": round(4 * insts / (busy * n_simd), 4) if busy else None}
print(json.dumps({"how": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace around tools/proof_counters.py; dispatches between "
                         "the two marker groups only; kernels are serialised by the counter collection, so busy_cycles are each kernel ALONE",
                  "proofs": P, "n_simd": n_simd, "n_xcd": n_xcd, "dispatches_counted": sum(int(a["launches"]) for a in acc.values()),
                  "valu_insts_per_proof": round(total_insts / P, 1), "issue_cycles_per_proof": round(4 * total_insts / P / n_simd, 1),
                  "serialised_busy_cycles_per_proof": round(total_busy / P, 1), "kernels": out}, indent=1))
