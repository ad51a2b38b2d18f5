#!/usr/bin/env python3
"""Kernel timeline of ONE warm proof out of a rocprofv3 --kernel-trace of tools/proof_counters.py (the dispatches between its two
marker launches), with the hardware queue of every dispatch: what ran beside what.
    python tools/trace_timeline.py <rocprof output dir> [proof index, default 1] [proofs between the markers, default 4]"""
import csv, glob, sys
d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "probe_inverse_kernel" in r["Kernel_Name"]]
# (a marker call launches the probe kernel several times: the proofs lie in the LARGEST gap between two marker dispatches)
if len(marks) >= 2:
    a = max(range(len(marks) - 1), key=lambda i: marks[i + 1] - marks[i])
    rows = rows[marks[a] + 1:marks[a + 1]]
def short(n):
    n = n.replace("void wsnark::", "").replace("wsnark::", "")
    base = n.split("(")[0]
    tag = "G2" if ("Fp2T" in base or "Fp2Pair" in base) else ("Fr" if "Fr29" in base or "FrParams" in base else "")
    return base.split("<")[0] + ("<" + tag + ">" if tag else "")
# a proof launches presort_count twice (witness plan, H plan); proofs are back to back, so cut at every second one
pc = [i for i, r in enumerate(rows) if "presort_count" in r["Kernel_Name"]]
starts = pc[0::2]
lo = starts[which] if which < len(starts) else 0
hi = starts[which + 1] if which + 1 < len(starts) else len(rows)
while lo > 0 and "rocclr" in rows[lo - 1]["Kernel_Name"]: lo -= 1
sel = rows[lo:hi]
qs = {}
t0 = int(sel[0]["Start_Timestamp"]); end = t0
ev = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    print("%9.1f us  +%8.1f  q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short(r["Kernel_Name"])))
    end = max(end, e); ev.append((s, 1)); ev.append((e, -1))
ev.sort(); depth = 0; last = t0; idle = 0
for t, dlt in ev:
    if depth == 0: idle += t - last
    depth += dlt; last = t
print("span %.1f us, kernels %d, time with NO kernel running %.1f us, sum of durations %.1f us" %
      ((end - t0) / 1e3, len(sel), idle / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e3))
