"""A/B of alternative BUILDS of libwsnark.so (compile-time switches): G1 / G2 MSM over 2^20 resident pairs with every
kernel bracketed, one process per build.  Development tool (not a test, not the bench):
    python tools/ab_msm.py label=path/to/libwsnark_variant.so [label2=...]
Alternative builds are made by tools/build_variant.sh into tools/alt/ (git-ignored .so files travel with gpurun)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path):
    import numpy as np
    import torch
    from wasmsnark_amd import _lib, bn128
    bn = bn128.Bn128(lib=_lib.load(path), device=0)
    n = 1 << 20
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
    d_s = torch.from_numpy(sc.reshape(-1)).cuda()
    out = {}
    for g, reps in ((1, 20), (2, 6)):
        pts = bn.mul_base(g, ks.tobytes())
        d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
        f = bn.g1_multiexp_dev if g == 1 else bn.g2_multiexp_dev
        torch.cuda.synchronize()
        for _ in range(3):
            r0 = f(d_s.data_ptr(), d_p.data_ptr(), n)
        t0 = time.perf_counter()
        for _ in range(reps):
            f(d_s.data_ptr(), d_p.data_ptr(), n)
        t = (time.perf_counter() - t0) / reps
        bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
        for _ in range(3):
            f(d_s.data_ptr(), d_p.data_ptr(), n)
        bn.lib.c.wsnark_timing_enable(0)
        out["g%d" % g] = {"ms": round(t * 1e3, 4), "result": r0[:8].hex(),
                          "kernel_ms": {k: round(v[0] / v[1], 4) for k, v in sorted(bn.lib.timing_report().items())}}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for spec in sys.argv[1:]:
            label, path = spec.split("=", 1)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(path)], capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
            print(json.dumps({"build": label}) [:-1] + ', "r": ' + line + "}")
