"""Post-processes a rocprofv3 --kernel-trace CSV of tools/trace_prove.py: prints the kernel timeline of the
LAST proof (name, start offset, duration, gap to the previous kernel) and totals."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void wsnark::", "").replace("wsnark::", "")
    base = n.split("(")[0]
    tag = "G2" if "Fp2T" in base else ("Fr" if "Fr29" in base or "FrParams" in base else "")
    return base.split("<")[0] + ("<" + tag + ">" if tag else "")
# a proof launches presort_count twice (witness plan first, then the H plan): the last proof starts at the
# second-to-last one, after the memsets that precede it
pc = [i for i, r in enumerate(rows) if "presort_count" in r["Kernel_Name"]]
idx = pc[-2] if len(pc) >= 2 else 0
while idx > 0 and "rocclr" in rows[idx - 1]["Kernel_Name"]: idx -= 1
sel = rows[idx:]
t0 = int(sel[0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(r["Kernel_Name"])))
    busy += e - s; prev_end = max(prev_end, e)
print("total span %.1f us, busy %.1f us, kernels %d" % ((prev_end - t0) / 1e3, busy / 1e3, len(sel)))
