#!/usr/bin/env python3
"""Per-kernel mean FETCH_SIZE / WRITE_SIZE from the two rocprofv3 --pmc passes of tools/gpu_session.sh
(counter_collection CSVs).  Prints JSON: {kernel: {"launches": n, "FETCH_SIZE": mean, "WRITE_SIZE": mean}}.
Units are the counters' own (KiB on gfx9 per rocprofv3's derived-metric definition); on gfx950
FETCH_SIZE under-reports wide coalesced streams 2x (MI355X_MICROARCH.md) -- raw values are kept here."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = defaultdict(lambda: {"launches": 0})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                if row.get("Counter_Name") != ctr:
                    continue
                short = name.split("(")[0].replace("void ", "").strip()[:70]
                try:
                    acc[short].append(float(row["Counter_Value"]))
                except (KeyError, ValueError):
                    pass
    for k, v in acc.items():
        out[k][ctr] = sum(v) / len(v)
        out[k]["launches"] = max(out[k]["launches"], len(v))
print(json.dumps({k: v for k, v in out.items() if "wsnark" in k or "radix" in k}, indent=1))
