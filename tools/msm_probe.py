"""Per-kernel times of one 2^k G1 sum: the per-call path (wsnark_g1_msm_dev) and the resident-bases path (wsnark_points_msm_dev,
fixed-base table plan).  Usage: python tools/msm_probe.py [--log-n 20] [--reps 30]; switches through WSNARK_* as usual."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--tag", default="")
    ap.add_argument("--lib", default=None, help="A/B: another build of the library (a subclass of the binding made here: the product's takes no path)")
    ap.add_argument("--set", action="append", default=[], help="NAME=value through wsnark_tuning_set (repeatable)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from wasmsnark_amd import bn128
    if a.lib:
        from wasmsnark_amd import _lib

        class Other(_lib.Lib):
            SO = os.path.abspath(a.lib)
        bn = bn128.Bn128(lib=Other())
    else:
        bn = bn128.build()
    for kv in a.set:
        k, v = kv.split("=")
        bn.lib.tune(k, int(v))
    n = 1 << a.log_n
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
    pts = bn.mul_base(1, ks.tobytes())
    d_s = torch.from_numpy(sc.reshape(-1)).cuda()
    d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    rp = bn.load_points(1, pts)
    out = {"tag": a.tag or ",".join(a.set) or "base", "log_n": a.log_n}
    ref = None
    for name, call in (("per_call", lambda: bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)), ("resident", lambda: rp.multiexp_dev(d_s.data_ptr(), n))):
        for _ in range(5):
            r = call()
        ref = ref or r
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.reps):
            call()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / a.reps
        bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
        for _ in range(a.reps):
            call()
        torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
        kt = bn.lib.timing_report()
        out[name] = {"ms": round(t * 1e3, 4), "same": r == ref,
                     "kernels_us": {k: round(v[0] / v[1] * 1e3, 1) for k, v in sorted(kt.items()) if v[1]}}
    print(json.dumps(out))


main()
