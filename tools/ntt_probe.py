"""NTT 2^22 (and 2^20) per direction with every pass bracketed: forward odd 0 / odd 1 / inverse, ms per transform and per pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wasmsnark_amd
bn = wasmsnark_amd.build(device=0)
for bits in (22, 20):
    m = 1 << bits
    x = torch.from_numpy(np.random.default_rng(7).integers(0, 256, size=(m, 32), dtype=np.uint8)); x[:, 31] &= 0x1F
    dx = x.reshape(-1).cuda()
    for odd, inv in ((0, False), (1, False), (0, True)):
        bn.fft_dev(dx.data_ptr(), m, odd, inverse=inv)
    bn.lib.c.wsnark_timing_report(None, 0)
    for name, odd, inv in (("fwd0", 0, False), ("inv", 0, True), ("fwd1", 1, False), ("fwd0", 0, False), ("inv", 0, True)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            bn.fft_dev(dx.data_ptr(), m, odd, inverse=inv)
        bn.lib.c.wsnark_timing_report(None, 0); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(1)
        for _ in range(3):
            bn.fft_dev(dx.data_ptr(), m, odd, inverse=inv)
        bn.lib.c.wsnark_timing_enable(0)
        k = {a: round(v[0] / v[1], 4) for a, v in bn.lib.timing_report().items()}
        print("2^%d %-5s %.4f ms/transform  kernels %s" % (bits, name, t * 1e3, k), flush=True)
