"""Runs the library's traffic-calibration kernels (wsnark_peak_probe 3 and 4: a known number of 64-byte gathers out of a
1 GiB table, and a 16-B-per-lane streaming read of it) and the integer-peak probes.  Under `rocprofv3 --pmc FETCH_SIZE`
the counter per launch of `probe_gather64_kernel` / `probe_stream16_kernel` against the known bytes is the calibration
of FETCH_SIZE for this repo's access patterns (tools/gpu_session.sh, summary in profiles/)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import wasmsnark_amd  # noqa: E402

bn = wasmsnark_amd.build(device=0)
out = {"known_bytes_per_launch": {"probe_gather64_kernel": (1 << 20) * 32 * 64, "probe_stream16_kernel": 1 << 30}, "launches_each": 3}
for name, probe in (("modmul_G_per_s", 0), ("modmul_inlined_G_per_s", 1), ("mad_u64_u32_G_per_s", 2), ("gather64_GB_per_s", 3), ("stream16_GB_per_s", 4), ("inversions_G_per_s", 5)):
    v = C.c_double(0)
    bn.lib.check(bn.lib.c.wsnark_peak_probe(probe, C.byref(v)))
    out[name] = round(v.value, 4 if v.value < 10 else 1)
out["products_per_inversion_equivalent"] = round(out["modmul_inlined_G_per_s"] / out["inversions_G_per_s"], 1) if out.get("inversions_G_per_s") else None
print(json.dumps(out))
