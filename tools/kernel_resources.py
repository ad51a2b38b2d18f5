"""Kernel-resource table of the shipped device code: VGPRs / AGPRs / SGPRs / scratch / spills / LDS / occupancy per kernel,
from the compiler's own remarks (hipcc -Rpass-analysis=kernel-resource-usage on every .hip source with the flags of
wasmsnark_amd/csrc/Makefile).  No GPU needed.  Usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.md"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wasmsnark_amd", "csrc")
SRCS = ["ntt.hip", "msm.hip", "calch.hip", "dist.hip", "fixedbase.hip", "selftest.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]
KEYS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def one(src):
    p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", "/dev/null"], cwd=CSRC, capture_output=True, text=True)
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"src": src, "name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z\[\]/ ]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return rows


def main():
    with ThreadPoolExecutor(len(SRCS)) as ex:
        rows = [r for rs in ex.map(one, SRCS) for r in rs]
    dm = demangle([r["name"] for r in rows])
    print("| source | kernel | " + " | ".join(KEYS) + " |")
    print("|---|---|" + "---|" * len(KEYS))
    for r in rows:
        name = dm.get(r["name"], r["name"])
        name = re.sub(r"\(.*$", "", name).replace("wsnark::", "").replace("void ", "")
        name = name.replace("Curve<Fp2T<Field29<Fq29Params> > >", "G2R29").replace("Curve<Field29I<Fq29Params> >", "G1R29I").replace("Curve<Field29<Fq29Params> >", "G1R29")
        name = name.replace("Curve<Fp2PairT<Field29<Fq29Params> > >", "G2P29").replace("Curve<Fp2T<Field<FqParams> > >", "G2").replace("Curve<Field<FqParams> >", "G1").replace("Field29<Fr29Params>", "Fr29").replace("Field<FrParams>", "Fr")
        print("| %s | `%s` | " % (r["src"], name) + " | ".join(r.get(k, "?") for k in KEYS) + " |")


if __name__ == "__main__":
    sys.exit(main())
