#!/usr/bin/env python3
"""Per-kernel mean FETCH_SIZE / WRITE_SIZE from the two rocprofv3 --pmc passes of tools/gpu_session.sh
(counter_collection CSVs) -> the JSON bench.py reads from profiles/r02_pmc_traffic.json:
  {"how": ..., "kernels": {short kernel name: {"launches": n, "fetch_bytes": mean per launch, "write_bytes": ...}}}
Counters are in KiB (rocprofv3's derived metrics on gfx9).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM):
on gfx950 FETCH_SIZE reports exactly half the bytes of wide coalesced streaming reads (128-B requests tallied at 64 B);
other access widths (the 64-B point gathers of msm_accumulate) and WRITE_SIZE are uncalibrated.  Both the raw value
and, for the streaming kernels listed in STREAMING, the doubled one are kept."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

STREAMING = ("ntt_pass_kernel", "fr_mul_kernel", "fr_map_kernel", "calch_combine_kernel", "presort_count", "msm_convert_points")
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = defaultdict(lambda: {"launches": 0})
for ctr, key in (("FETCH_SIZE", "fetch_bytes"), ("WRITE_SIZE", "write_bytes")):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                if row.get("Counter_Name") != ctr:
                    continue
                m = re.search(r"wsnark::([A-Za-z0-9_]+)", name)
                short = m.group(1) if m else name.split("(")[0].replace("void ", "").strip()[:60]
                # distinguish the G1 / G2 instantiations of the templated MSM kernels
                if m and ("Fp2" in name or "Fe2T" in name):
                    short += "_g2"
                try:
                    acc[short].append(float(row["Counter_Value"]) * 1024.0)
                except (KeyError, ValueError):
                    pass
    for k, v in acc.items():
        out[k][key] = int(sum(v) / len(v))
        out[k]["launches"] = max(out[k]["launches"], len(v))
kern = {}
for k, v in sorted(out.items()):
    v.setdefault("fetch_bytes", 0); v.setdefault("write_bytes", 0)
    if any(k.startswith(s) for s in STREAMING):
        v["fetch_bytes_corrected_x2"] = 2 * v["fetch_bytes"]
    kern[k] = v
# bench.py keys: the timer's names
alias = {"msm_accumulate_g1": "msm_accumulate", "msm_accumulate_g2": "msm_accumulate_g2"}
for a, k in alias.items():
    if k in kern:
        kern[a] = kern[k]
# the shape of the msm_accumulate launches the means were taken over: from the bench line of the same pass
shape = None
try:
    for line in open(os.path.join(root, "pmc_FETCH_SIZE.log")):
        if line.startswith("{") and '"roofline_int_alu"' in line:
            d = json.loads(line)
            shape = {"pairs_per_launch": d["roofline"]["algorithmic_bytes_per_launch"] // 96,
                     "windows_per_launch": d["roofline_int_alu"].get("windows_per_launch")}
except Exception:  # noqa: BLE001
    pass
print(json.dumps({"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (--kernel-trace only) around "
                         "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` (proofs only: every msm_accumulate launch has "
                         "the shape below); mean bytes per launch = counter (KiB) x 1024; "
                         "FETCH_SIZE raw except *_corrected_x2 (gfx950 tallies wide coalesced 128-B requests at 64 B)",
                  "msm_accumulate_shape": shape, "kernels": kern}, indent=1))
