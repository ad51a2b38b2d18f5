"""Generates the small synthetic proving keys / witnesses under tests/golden/keys/ (committed).
Run once in the build container, followed by `node oracle/ref_harness/gen_golden.js proofs`,
which makes the REFERENCE prove on these keys (fixed r, s) and verify its own proofs
(tests/golden/proofs.json).  Group arithmetic here uses the pinned CPU oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402
from wasmsnark_amd import synth  # noqa: E402
from wasmsnark_amd.bn128 import G1_GEN, G2_GEN  # noqa: E402


def oracle_mul_base(g, scalars):
    sz = 64 if g == 1 else 128
    one = orc.f_un("to_mont", 0, (1).to_bytes(32, "little"))
    gen = (G1_GEN + one) if g == 1 else (G2_GEN + one + b"\0" * 32)
    out = bytearray()
    for i in range(0, len(scalars), 32):
        p = orc.g_affine(g, orc.g_times_scalar(g, gen, scalars[i:i + 32]))
        out += b"\0" * sz if orc.g_is_zero(g, p) else p[:sz]
    return bytes(out)


def main():
    out = os.path.join(ROOT, "tests", "golden", "keys")
    os.makedirs(out, exist_ok=True)
    for name, logd, npub, seed in (("t3", 3, 1, 11), ("t6", 6, 3, 12)):
        circ = synth.make_circuit(logd, n_public=npub, seed=seed, style="rows")   # (the committed keys: round-1 generator)
        S = synth.setup(circ, seed=seed + 100)
        pkey, vk = synth.build_key(circ, S, oracle_mul_base)
        open(os.path.join(out, name + ".pkey.bin"), "wb").write(pkey)
        open(os.path.join(out, name + ".witness.bin"), "wb").write(synth.witness_bin(circ))
        json.dump(vk, open(os.path.join(out, name + ".vk.json"), "w"))
        json.dump(synth.public_signals(circ), open(os.path.join(out, name + ".public.json"), "w"))
        json.dump({"log_domain": logd, "n_public": npub, "circuit_seed": seed, "setup_seed": seed + 100},
                  open(os.path.join(out, name + ".meta.json"), "w"))
        print(name, "nVars", circ.n_vars, "domain", circ.domain, "key bytes", len(pkey))


if __name__ == "__main__":
    main()
