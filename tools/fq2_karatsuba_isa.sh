#!/bin/bash
# ISA instruction counts of the two Fq2 product forms (tools/fq2_karatsuba_isa.hip), gfx950, hipcc -O3.  No GPU needed.
cd "$(dirname "$0")/.." || exit 1
S=/tmp/fq2_karatsuba_isa.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Iwasmsnark_amd/csrc tools/fq2_karatsuba_isa.hip -o $S || exit 1
python3 - "$S" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for name in ("k_shipped", "k_karatsuba"):
    m = re.search(r"^_ZN6wsnark%d%sE\w*:[^\n]*\n(.*?)s_endpgm" % (len(name), name), txt, re.S | re.M)
    body = m.group(1)
    ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";")) and l.split()]
    valu = [i for i in ins if i.startswith("v_")]
    mad = [i for i in valu if i.startswith("v_mad_u64_u32") or i.startswith("v_mad_i64_i32")]
    mem = [i for i in ins if i.startswith(("global_", "buffer_", "flat_", "scratch_"))]
    vg = re.search(r"_ZN6wsnark%d%sE\w*\.num_vgpr, (\d+)" % (len(name), name), txt)
    print("%-12s VALU %4d  of which 64-bit multiply-adds %4d, other VALU %4d;  all instructions %4d (memory %d);  VGPRs %s"
          % (name, len(valu), len(mad), len(valu) - len(mad), len(ins), len(mem), vg.group(1) if vg else "?"))
PY
