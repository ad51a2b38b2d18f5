"""Prover timing with a boolean-heavy witness (80 % of the signals are 0 or 1, as in bit-decomposition circuits).
The prover is a pure function of (witness, key, r, s): validity of the witness does not matter for timing, and
the result is cross-checked between the one-queue and the two-queue schedule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
circ = synth.make_circuit(20, n_public=5, seed=1); S = synth.setup(circ, seed=2)
pkey, _ = synth.build_key(circ, S, bn.mul_base); key = bn.load_key(pkey)
n = circ.n_vars
rng = np.random.default_rng(3)
w = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); w[:, 31] &= 0x1F
mask = rng.random(n) < 0.8
w[mask] = 0; w[mask, 0] = rng.integers(0, 2, size=int(mask.sum()))
w[0] = 0; w[0, 0] = 1
for name, arr in (("uniform (valid witness)", np.frombuffer(synth.witness_bin(circ), dtype=np.uint8).reshape(n, 32)), ("80% boolean", w)):
    d_w = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1)).cuda(); torch.cuda.synchronize()
    r32, s32 = bytes(range(32)), bytes(range(32, 64))
    ref = bn.groth16GenProof_dev(d_w.data_ptr(), n * 32, key, r=r32, s=s32)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = bn.groth16GenProof_dev(d_w.data_ptr(), n * 32, key, r=r32, s=s32)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert p == ref
    print("%-26s prove ms: %s" % (name, " ".join("%.2f" % t for t in ts)))
