"""A/B of the accumulation SHAPE on one stand-alone G1 MSM of 2^20 pairs: the shipped kernel (one lane =
one bucket run, in registers) against WSNARK_ACC_SHAPE=segscan (one pair per lane, wavefront segmented scan staged through
LDS over the bucket-sorted stream) -- uniform scalars and the circuit-like histogram of SURVEY.md section 8d (6.7 % zeros,
3.1 % ones, 10 % below 2^32).  The environment variable is read per launch: both shapes run in this one process on the same
inputs and must return the same sums."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import wasmsnark_amd
bn = wasmsnark_amd.build(device=0)
n = 1 << 20
rng = np.random.default_rng(1234)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
u = rng.random(n)
sk = sc.copy(); sk[u < 0.067] = 0
ones = (u >= 0.067) & (u < 0.098); sk[ones] = 0; sk[ones, 0] = 1
small = (u >= 0.098) & (u < 0.2); sk[small, 4:] = 0
d_p = torch.frombuffer(bytearray(bn.mul_base(1, ks.tobytes())), dtype=torch.uint8).cuda()
out = {}
reps = int(os.environ.get("AB_REPS", "10"))
for dist, arr in (("uniform", sc), ("circuit_like", sk)):
    d_s = torch.from_numpy(arr.reshape(-1)).cuda(); torch.cuda.synchronize()
    res = {}
    for shape in ("lane_per_bucket", "segscan"):
        if shape == "segscan":
            os.environ["WSNARK_ACC_SHAPE"] = "segscan"
        else:
            os.environ.pop("WSNARK_ACC_SHAPE", None)
        f = lambda: bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
        for _ in range(3):
            r0 = f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
        bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
        for _ in range(3):
            f()
        torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
        kt = {k: round(v[0] / v[1], 4) for k, v in sorted(bn.lib.timing_report().items())}
        res[shape] = {"msm_ms": round(t * 1e3, 3), "accumulate_ms": kt.get("msm_accumulate_g1"), "combine_or_merge_ms": kt.get("msm_combine"), "result": r0.hex()[:16]}
    res["same_result"] = res["lane_per_bucket"]["result"] == res["segscan"]["result"]
    out[dist] = res
os.environ.pop("WSNARK_ACC_SHAPE", None)
print(json.dumps(out))
