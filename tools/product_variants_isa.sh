#!/bin/bash
# ISA counts by issue class of the shipped radix-2^29 product and its Karatsuba 5+4 variant (tools/product_variants_isa.hip), gfx950, hipcc -O3,
# priced with the class rates of tools/issue_model.py (DEFAULT_RATES = MI355X, gpurun call r06_c05, or --rates <issue_classes.json>).  No GPU needed.
cd "$(dirname "$0")/.." || exit 1
S=/tmp/product_variants_isa.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -DWS_USE_MAD_CHAIN_OFF -Iwasmsnark_amd/csrc tools/product_variants_isa.hip -o $S 2>/dev/null || exit 1
python3 - "$S" "$@" <<'PY'
import json, os, re, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import isa_histogram as ih, issue_model as im
rates = dict(im.DEFAULT_RATES)
if "--rates" in sys.argv:
    rates.update(json.load(open(sys.argv[sys.argv.index("--rates") + 1]))["G_lane_ops_per_s"])
txt = open(sys.argv[1]).read()
cyc, clock = im.equivalent_cycles(rates)
print("class costs (cycles per wave-instruction at the %.3f GHz that makes v_add_u32 2 cycles): " % clock + ", ".join("%s %.2f" % (c, cyc[ih.PROBE_OF[c]]) for c in ih.CLASSES if c != "select_run"))
for name in ("k_shipped", "k_karatsuba"):
    m = re.search(r"^_ZN6wsnark%d%sE\w*:[^\n]*\n(.*?)s_endpgm" % (len(name), name), txt, re.S | re.M)
    lines = [l for l in m.group(1).splitlines() if l.startswith("\t") and l.split() and not l.strip().startswith((".", ";"))]
    h = dict.fromkeys(ih.CLASSES, 0)
    for l in lines:
        if l.split()[0].startswith("v_"):
            h[ih.classify(l)] += 1
    n = sum(h.values())
    t = sum(v * cyc[ih.PROBE_OF[c]] for c, v in h.items())
    print("%-12s VALU %4d  " % (name, n) + "  ".join("%s %d" % (c, h[c]) for c in ih.CLASSES if h[c]) + "   -> %.0f cycles per product and wavefront" % t)
PY
