"""Witness values and key coefficients in [r, 2^256): inputs the reference accepts and nothing canonicalises before the prover sees
them (fft_toMontgomeryN reduces the signals, src/build_fft.js:418-458; the multiexps take raw 256-bit scalars,
src/build_multiexp.js:651-744; pol_constructLC multiplies whatever the key holds, src/build_pol.js:62-144).
tests/golden/unreduced.json holds what the REFERENCE ITSELF computes for them (oracle/ref_harness/gen_golden.js unreduced): CALC_H
instances and whole proofs on the t6 key with the witness, the key's coefficients, and both, lifted by multiples of r.
CPU: the oracle and the kernel sources under the thread emulator; -m gpu: the product library, through ctypes and through the
Node.js addon."""
import base64
import json
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT, load_golden

B64 = base64.b64decode


def _check(bn):
    U = load_golden("unreduced.json")
    for c in U["calch"]:
        assert bn.calcH(B64(c["signals"]), B64(c["polsA"]), B64(c["polsB"]), c["nSignals"], c["domain"]) == B64(c["h"])
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", U["key"] + ext), "rb").read()
    wit, pkey = rd(".witness.bin"), rd(".pkey.bin")
    wit2, pkey2 = B64(U["witness_lifted"]), B64(U["pkey_lifted"])
    assert wit2 != wit and pkey2 != pkey and U["coefficients_lifted"] > 0
    inputs = {"witness lifted": (wit2, pkey), "coefficients lifted": (wit, pkey2), "both lifted": (wit2, pkey2)}
    canon = {(c["r"], c["s"]): c["proof"] for c in load_golden("proofs.json")[U["key"]]}
    keys = {}
    for c in U["proofs"]:
        assert c["reference_verifies"] is True
        w, k = inputs[c["label"]]
        if id(k) not in keys:
            keys[id(k)] = bn.load_key(k)
        got = bn.groth16GenProof(w, keys[id(k)], r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"]))
        assert got == c["proof"], c["label"]
        # a lifted value is the same field element: the reference's proof is the one it gives on the canonical inputs
        assert c["proof"] == canon[(c["r"], c["s"])]
    for k in keys.values():
        k.free()


def test_oracle_matches_the_reference_on_unreduced_inputs(orc):
    U = load_golden("unreduced.json")
    for c in U["calch"]:
        assert orc.calc_h(B64(c["signals"]), B64(c["polsA"]), B64(c["polsB"]), c["nSignals"], c["domain"]) == B64(c["h"])
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", U["key"] + ext), "rb").read()
    inputs = {"witness lifted": (B64(U["witness_lifted"]), rd(".pkey.bin")), "coefficients lifted": (rd(".witness.bin"), B64(U["pkey_lifted"])),
              "both lifted": (B64(U["witness_lifted"]), B64(U["pkey_lifted"]))}
    for c in U["proofs"]:
        w, k = inputs[c["label"]]
        assert orc.groth16_prove(w, k, bytes.fromhex(c["r"]), bytes.fromhex(c["s"]), workers=8) == c["proof"], c["label"]


def test_emulated_kernels_on_unreduced_inputs():
    from emul_util import emul_bn128
    _check(emul_bn128())


@pytest.mark.gpu
def test_gpu_on_unreduced_inputs():
    import wasmsnark_amd
    _check(wasmsnark_amd.build(device=0))


NODE_SCRIPT = r"""
const fs = require("fs"), path = require("path");
const root = process.argv[1];                                       // (node -e: the arguments start at argv[1])
if (process.argv[2]) require(path.join(root, "tests", "emul", "use_emulator_addon.js"));   // tests only: the emulator build of the addon
const ws = require(path.join(root, "wasmsnark_amd", "js", "index.js"));
const gold = path.join(root, "tests", "golden");
(async () => {
    const U = JSON.parse(fs.readFileSync(path.join(gold, "unreduced.json"), "utf8"));
    const bn = await ws.buildBn128();
    for (const c of U.calch) {
        const h = await bn.calcH(Buffer.from(c.signals, "base64"), Buffer.from(c.polsA, "base64"), Buffer.from(c.polsB, "base64"), c.nSignals, c.domain);
        if (Buffer.from(h).toString("base64") !== c.h) throw new Error("calcH on unreduced inputs");
    }
    const rd = (ext) => fs.readFileSync(path.join(gold, "keys", U.key + ext));
    const inputs = { "witness lifted": [Buffer.from(U.witness_lifted, "base64"), rd(".pkey.bin")], "coefficients lifted": [rd(".witness.bin"), Buffer.from(U.pkey_lifted, "base64")],
                     "both lifted": [Buffer.from(U.witness_lifted, "base64"), Buffer.from(U.pkey_lifted, "base64")] };
    for (const c of U.proofs) {
        const [w, k] = inputs[c.label];
        const p = await bn.groth16GenProof(w, k, { r: Buffer.from(c.r, "hex"), s: Buffer.from(c.s, "hex") });
        if (JSON.stringify(p) !== JSON.stringify(c.proof)) throw new Error("proof on unreduced inputs: " + c.label);
    }
    console.log("NODE_UNREDUCED_OK");
    bn.terminate();
})().catch((e) => { console.error("NODE_UNREDUCED_FAIL", e); process.exit(1); });
"""


def _node(lib=None):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "wasmsnark_amd", "js"), "-s", "all"] + (["emul"] if lib else []))
    return subprocess.run(["node", "-e", NODE_SCRIPT, ROOT] + ([lib] if lib else []), capture_output=True, text=True, timeout=600)


needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"), reason="node / N-API headers not available")


@needs_node
def test_node_addon_on_unreduced_inputs_emulated():
    from emul_util import emul_bn128, SO
    emul_bn128()
    out = _node(SO)
    assert out.returncode == 0 and "NODE_UNREDUCED_OK" in out.stdout, out.stdout + out.stderr


@needs_node
@pytest.mark.gpu
def test_node_addon_on_unreduced_inputs_gpu():
    out = _node()
    assert out.returncode == 0 and "NODE_UNREDUCED_OK" in out.stdout, out.stdout + out.stderr
