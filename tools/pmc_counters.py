#!/usr/bin/env python3
"""Per-kernel means of every counter found in the rocprofv3 --pmc passes under a directory (counter_collection CSVs) ->
JSON {kernel: {"launches": n, counter: mean per launch, ...}} plus, where the SQ / GRBM counters are present, the ratios
DESIGN.md quotes:
  valu_inst_per_wave        SQ_INSTS_VALU / SQ_WAVES
  issue_floor_frac          (round 6) the kernel's issue floor by INSTRUCTION CLASS over its measured duration: SQ_INSTS_VALU x 64 x
                            sum(class share / class rate) / mean launch duration (tools/issue_model.py: static class mix from
                            profiles/rNN_isa_classes.json, class rates measured on the box -- --rates <issue_classes.json>; durations from
                            the kernel trace of the same pass).  Never above 1 for a kernel whose mix is known; `issue_floor_how` says
                            what priced it.
  valu_slots_flat4          4 x SQ_INSTS_VALU / ((GRBM_GUI_ACTIVE / n_xcd) x n_simd): the round-3..5 figure ("valu_issue_frac" then),
                            every VALU instruction priced at 4 cycles.  WRONG as a utilisation -- plain 32-bit instructions issue in 2
                            (mul_base_kernel read 1.34) -- kept only so that old and new records can be compared.  GRBM_GUI_ACTIVE
                            comes back summed over the 8 XCDs; on gfx950 SQ_ACTIVE_INST_VALU returns the same number as SQ_INSTS_VALU.
  wave_valu_frac            SQ_INSTS_VALU / SQ_WAVE_CYCLES
  wait_inst_frac, wait_any_frac   SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES
Usage: tools/pmc_counters.py <dir with pmc_* subdirectories> [n_simd [n_xcd]] [--rates issue_classes.json] [--classes isa_classes.json]"""
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
root = argv[0] if argv else "gpurun_out"
n_simd = int(argv[1]) if len(argv) > 1 else 1024
n_xcd = int(argv[2]) if len(argv) > 2 else 8
classes = issue_model.load_classes(opt.get("--classes"))
rates, rates_src = dict(issue_model.DEFAULT_RATES), "tools/issue_model.py DEFAULT_RATES (gpurun call r06_c05)"
if opt.get("--rates") and os.path.exists(opt["--rates"]):
    try:
        rates.update(json.load(open(opt["--rates"]))["G_lane_ops_per_s"])
        rates_src = opt["--rates"]
    except Exception:  # noqa: BLE001
        pass


def short_name(name):
    m = re.search(r"wsnark::([A-Za-z0-9_]+)", name)
    short = m.group(1) if m else name.split("(")[0].replace("void ", "").strip()[:60]
    if m and ("Fp2" in name or "Fe2T" in name):
        short += "_g2"
    return short


acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)          # kernel -> launch durations (ns), from the kernel traces of the same passes
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            short = short_name(row.get("Kernel_Name") or row.get("Kernel Name") or "")
            try:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            except (KeyError, ValueError):
                pass
            d = row.get("Dispatch_Id")
            if row.get("Start_Timestamp") and row.get("End_Timestamp") and d not in seen:
                seen.add(d)
                try:
                    dur[short].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                except ValueError:
                    pass
if not dur:
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                try:
                    dur[short_name(row.get("Kernel_Name") or "")].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
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
        d["valu_slots_flat4"] = round(4 * d["SQ_INSTS_VALU"] / (d["GRBM_GUI_ACTIVE"] / n_xcd * n_simd), 4)
    if g("SQ_INSTS_VALU") and dur.get(k):
        mean_s = sum(dur[k]) / len(dur[k]) / 1e9
        dyn64 = d["SQ_INSTS_VALU_INT64"] / d["SQ_INSTS_VALU"] if g("SQ_INSTS_VALU_INT64") is not None else None
        floor_s, how = issue_model.kernel_floor_s(k, d["SQ_INSTS_VALU"], classes, rates, dyn64)
        d["mean_launch_us"] = round(mean_s * 1e6, 2)
        d["issue_floor_us"] = round(floor_s * 1e6, 2)
        d["issue_floor_frac"] = round(floor_s / mean_s, 4) if mean_s > 0 else None
        d["issue_floor_how"] = how
    if g("SQ_INSTS_VALU") and g("SQ_WAVE_CYCLES"):
        d["wave_valu_frac"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVE_CYCLES"], 4)
    if g("SQ_WAVE_CYCLES"):
        for c, r in (("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_frac")):
            if g(c) is not None:
                d[r] = round(d[c] / d["SQ_WAVE_CYCLES"], 4)
    out[k] = d
print(json.dumps({"how": "rocprofv3 --pmc <counters> --kernel-trace, one pass per counter group (tools/gpu_session.sh); means per launch; "
                         "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; issue_floor_frac: tools/issue_model.py", "n_simd": n_simd, "n_xcd": n_xcd,
                  "class_rates_G_lane_ops_per_s": rates, "class_rates_source": rates_src, "class_mix_source": os.path.basename(classes["_path"]) if classes else None,
                  "kernels": out}, indent=1))
