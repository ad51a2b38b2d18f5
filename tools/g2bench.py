import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wasmsnark_amd
bn = wasmsnark_amd.build(device=0)
n = 1 << 20
rng = np.random.default_rng(5)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
pts = bn.mul_base(2, ks.tobytes())
d_s = torch.from_numpy(sc.reshape(-1)).cuda(); d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for _ in range(2): bn.g2_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
t0 = time.perf_counter()
for _ in range(5): bn.g2_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
t = (time.perf_counter() - t0) / 5
print("g2 msm 2^20 ms", round(t * 1e3, 3), {k: round(v[0] / v[1], 3) for k, v in bn.lib.timing_report().items()})
