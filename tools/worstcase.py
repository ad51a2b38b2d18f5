"""Degenerate scalar distributions: how far the 2^20 G1 MSM degrades (it must stay exact; see the GPU tests)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wasmsnark_amd
bn = wasmsnark_amd.build(device=0)
n = 1 << 20
rng = np.random.default_rng(5)
ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
d_p = torch.frombuffer(bytearray(bn.mul_base(1, ks.tobytes())), dtype=torch.uint8).cuda()
def t(sc, reps=5):
    d_s = torch.from_numpy(sc.reshape(-1)).cuda(); torch.cuda.synchronize()
    bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(reps): bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
    dt = (time.perf_counter() - t0) / reps * 1e3
    bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
    bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
    bn.lib.c.wsnark_timing_enable(0)
    print("   ", {k.replace("msm_", ""): round(v[0], 3) for k, v in bn.lib.timing_report().items()})
    return dt
uni = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); uni[:, 31] &= 0x1F
same = np.tile(uni[0], (n, 1))
two = uni.copy(); two[:] = uni[0]; two[::2] = uni[1]
small = np.zeros((n, 32), dtype=np.uint8); small[:, 0] = rng.integers(0, 4, size=n)
ones = np.zeros((n, 32), dtype=np.uint8); ones[:, 0] = 1
bits = np.zeros((n, 32), dtype=np.uint8); bits[:, 0] = rng.integers(0, 2, size=n)      # boolean wires
mixed = uni.copy(); mask = rng.random(n) < 0.8; mixed[mask] = 0; mixed[mask, 0] = rng.integers(0, 2, size=int(mask.sum()))
print("uniform            %.3f ms" % t(uni))
print("all scalars equal  %.3f ms" % t(same))
print("two distinct values %.3f ms" % t(two))
print("values 0..3        %.3f ms" % t(small))
print("all ones           %.3f ms" % t(ones))
print("booleans (0/1)     %.3f ms" % t(bits))
print("80%% booleans + 20%% uniform %.3f ms" % t(mixed))
