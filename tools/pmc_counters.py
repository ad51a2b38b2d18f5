#!/usr/bin/env python3
"""Per-kernel means of every counter found in the rocprofv3 --pmc passes under a directory (counter_collection CSVs) ->
JSON {kernel: {"launches": n, counter: mean per launch, ...}} plus, where the SQ / GRBM counters are present, the ratios
DESIGN.md quotes:
  valu_inst_per_wave        SQ_INSTS_VALU / SQ_WAVES
  valu_issue_frac           4 x SQ_INSTS_VALU / ((GRBM_GUI_ACTIVE / n_xcd) x n_simd): the share of the chip's VALU issue slots the
                            launch used.  A wave64 VALU instruction occupies its SIMD16 for 4 cycles, so one SIMD issues at most one
                            wave-instruction per 4 cycles; GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (a 1 ms launch reads
                            ~19.6 M = 8 x 2.45 M cycles), SQ_INSTS_VALU summed over all waves.  On gfx950 SQ_ACTIVE_INST_VALU
                            returns the same number as SQ_INSTS_VALU (an instruction count, not quad-cycles).  Cross-check: the
                            pure product chain of wsnark_peak_probe reads 0.98.
  wave_valu_frac            SQ_INSTS_VALU / SQ_WAVE_CYCLES
  wait_inst_frac, wait_any_frac   SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES
Usage: tools/pmc_counters.py <dir with pmc_* subdirectories> [n_simd]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
n_simd = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n_xcd = int(sys.argv[3]) if len(sys.argv) > 3 else 8
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            m = re.search(r"wsnark::([A-Za-z0-9_]+)", name)
            short = m.group(1) if m else name.split("(")[0].replace("void ", "").strip()[:60]
            if m and ("Fp2" in name or "Fe2T" in name):
                short += "_g2"
            try:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            except (KeyError, ValueError):
                pass
out = {}
for k, ctrs in sorted(acc.items()):
    d = {"launches": max(len(v) for v in ctrs.values())}
    for c, v in sorted(ctrs.items()):
        d[c] = round(sum(v) / len(v), 1)
    g = d.get
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        d["valu_inst_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
        d["valu_issue_frac"] = round(4 * d["SQ_INSTS_VALU"] / (d["GRBM_GUI_ACTIVE"] / n_xcd * n_simd), 4)
    if g("SQ_INSTS_VALU") and g("SQ_WAVE_CYCLES"):
        d["wave_valu_frac"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVE_CYCLES"], 4)
    if g("SQ_WAVE_CYCLES"):
        for c, r in (("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_frac")):
            if g(c) is not None:
                d[r] = round(d[c] / d["SQ_WAVE_CYCLES"], 4)
    out[k] = d
print(json.dumps({"how": "rocprofv3 --pmc <counters> --kernel-trace, one pass per counter group (tools/gpu_session.sh); means per launch; "
                         "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them", "n_simd": n_simd, "n_xcd": n_xcd, "kernels": out}, indent=1))
