"""A/B of alternative BUILDS of libwsnark.so on whole proofs (2^20, the bench's circuit): one process per build.
    python tools/ab_prove.py label=path/to/libwsnark_variant.so [label2=...]      (builds: tools/build_variant.sh)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, logd):
    import torch
    from wasmsnark_amd import _lib, bn128, synth
    bn = bn128.Bn128(lib=_lib.load(path), device=0)
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
    sec, _ = circ.build_sections()
    key = bn.load_key(sections=sec)
    wit = circ.witness_bin()
    d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    r, s = bytes(range(32)), bytes(range(32, 64))
    want = circ.expected_proof(r, s)
    for _ in range(5):
        p = bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
    best = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 20 * 1e3)
    bn.lib.c.wsnark_timing_reset(); bn.lib.c.wsnark_timing_enable(2)
    for _ in range(5):
        bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
    torch.cuda.synchronize(); bn.lib.c.wsnark_timing_enable(0)
    kt = {k: round(v[0] / v[1], 4) for k, v in sorted(bn.lib.timing_report().items())}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        bn.groth16GenProof(wit, key, r=r, s=s)
    host = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"prove_ms": [round(x, 3) for x in best], "host_witness_ms": round(host, 3), "ok": p == want, "acc_launch_ms": kt,
                      "key_load_ms": {k: round(v, 1) for k, v in key.load_ms.items()}}))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    else:
        for spec in sys.argv[1:]:
            label, path = spec.split("=", 1)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(path)], capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:]
            print(json.dumps({"build": label})[:-1] + ', "r": ' + line + "}")
