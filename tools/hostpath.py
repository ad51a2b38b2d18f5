"""Host-pointer boundary timing (what a Node / ctypes caller with plain host buffers sees):
G1 MSM 2^20 through wsnark_g1_msm vs the device-pointer entry, and prove with a host witness."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wasmsnark_amd
bn = wasmsnark_amd.build(device=0)
n = 1 << 20
rng = np.random.default_rng(5)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
pts = bn.mul_base(1, ks.tobytes())
scb = sc.tobytes()
d_s = torch.from_numpy(sc.reshape(-1)).cuda(); d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
def t(f, reps=8):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e3
a = t(lambda: bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n))
b = t(lambda: bn.g1_multiexp(scb, pts))
assert bn.g1_multiexp(scb, pts) == bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
print("g1 msm 2^20: device pointers %.3f ms, host pointers %.3f ms (96 MB H2D inside)" % (a, b))
# fresh host pages every call (what a caller that builds new buffers per request looks like)
ts = []
for i in range(5):
    s2 = bytes(bytearray(scb)); p2 = bytes(bytearray(pts))
    t0 = time.perf_counter(); bn.g1_multiexp(s2, p2); ts.append((time.perf_counter() - t0) * 1e3)
print("host pointers, fresh buffers each call: " + " ".join("%.2f" % x for x in ts) + " ms")
from wasmsnark_amd import synth
circ = synth.make_circuit(18, n_public=5, seed=1); S = synth.setup(circ, seed=2)
pkey, _ = synth.build_key(circ, S, bn.mul_base)
t0 = time.perf_counter(); key = bn.load_key(pkey); print("load_key 2^18 (%.0f MB): %.1f ms" % (len(pkey) / 1e6, (time.perf_counter() - t0) * 1e3))
wit = synth.witness_bin(circ)
r32, s32 = bytes(range(32)), bytes(range(32, 64))
assert bn.groth16GenProof(wit, key, r=r32, s=s32) == synth.expected_proof(circ, S, r32, s32, bn.mul_base)
print("prove 2^18 from a host witness: %.3f ms" % t(lambda: bn.groth16GenProof(wit, key, r=r32, s=s32), 5))
