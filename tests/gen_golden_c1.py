"""BASELINE config 1: the reference's own example witness (example/bn128/witness.bin, 66 232 signals,
58 public) proved by the REFERENCE (Node + WASM, 8 workers) on a seeded pseudo-key
(wasmsnark_amd.synth.pseudo_key; the real proving key is missing from the reference checkout).
Writes tests/golden/c1_example.json (seed, r, s, the reference's proof) and copies the witness data
file to tests/golden/example_witness.bin.  Run once in the build container."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gen_golden_keys import oracle_mul_base  # noqa: E402
from wasmsnark_amd import synth  # noqa: E402

REF = os.environ.get("WSNARK_REF", "/root/reference")
N_VARS, N_PUBLIC, DOMAIN, SEED = 66232, 58, 131072, 20260926

NODE = r'''
const E = require(process.argv[2] + "/oracle/ref_harness/refenv.js");
const fs = require("fs");
(async () => {
  const { bn } = await E.buildRef();
  const pkey = fs.readFileSync(process.argv[3]), wit = fs.readFileSync(process.argv[4]);
  const r = Buffer.from(process.argv[5], "hex"), s = Buffer.from(process.argv[6], "hex");
  E.setRS(r, s);
  const t0 = Date.now();
  const proof = await bn.groth16GenProof(E.toAB(new Uint8Array(wit)), E.toAB(new Uint8Array(pkey)));
  const ms = Date.now() - t0;
  const used = [Buffer.from(bn.getBin(bn._pr, 32)).toString("hex"), Buffer.from(bn.getBin(bn._ps, 32)).toString("hex")];
  console.log(JSON.stringify({ proof, used, ms }));
  bn.terminate();
})().catch((e) => { console.error(e); process.exit(1); });
'''


def main():
    wit_src = os.path.join(REF, "example", "bn128", "witness.bin")
    wit_dst = os.path.join(ROOT, "tests", "golden", "example_witness.bin")
    shutil.copyfile(wit_src, wit_dst)
    assert os.path.getsize(wit_dst) == N_VARS * 32
    pkey = synth.pseudo_key(N_VARS, N_PUBLIC, DOMAIN, SEED, oracle_mul_base)
    r = bytes(range(1, 33)).hex()
    s = (b"\xff" * 31 + b"\x7f").hex()
    with tempfile.TemporaryDirectory() as td:
        kp = os.path.join(td, "pkey.bin")
        open(kp, "wb").write(pkey)
        js = os.path.join(td, "run.js")
        open(js, "w").write(NODE)
        out = subprocess.check_output(["node", js, ROOT, kp, wit_dst, r, s], text=True)
    res = json.loads(out.strip().splitlines()[-1])
    assert res["used"] == [r, s]
    json.dump({"n_vars": N_VARS, "n_public": N_PUBLIC, "domain": DOMAIN, "key_seed": SEED, "r": r, "s": s,
               "proof": res["proof"], "reference_prove_ms_8_workers_build_container": res["ms"],
               "pkey_sha256": __import__("hashlib").sha256(pkey).hexdigest()},
              open(os.path.join(ROOT, "tests", "golden", "c1_example.json"), "w"), indent=1)
    print("reference prove took", res["ms"], "ms; key bytes", len(pkey))


if __name__ == "__main__":
    main()
