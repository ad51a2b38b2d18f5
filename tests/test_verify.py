"""Native Groth16 verification (wsnark_groth16_verify, host arithmetic: SURVEY.md section 8f row 4) against the reference
verifier's verdicts (Bn128.groth16Verify, /root/reference src/bn128.js:722-791):
  * tests/golden/verify.json: the reference's own verifier data (example/bn128 + test/data: a real 58-input circuit, three
    shipped proofs) and tampered variants, verdicts recorded from the reference by oracle/ref_harness/gen_verify_golden.js;
  * tests/golden/proofs.json + keys/*.vk.json: every golden proof of the two synthetic keys (the reference accepted them and
    rejected a wrong public input);
  * the two hosts of the entry point: the product library itself (host-only code: loads and runs without a GPU) and the
    thread-emulator build of the same sources."""
import json
import os

import pytest

from conftest import GOLDEN, ROOT, load_golden

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module", params=["product", "emul"])
def lib(request):
    from wasmsnark_amd import _lib
    if request.param == "product":
        so = os.path.join(ROOT, "wasmsnark_amd", "libwsnark.so")
        if not os.path.exists(so):
            import __graft_entry__
            __graft_entry__.build()
        return _lib.Lib(so)            # no init(): verification needs no GPU
    from emul_util import emul_bn128
    return emul_bn128().lib


def test_reference_verifier_vectors(lib):
    from wasmsnark_amd.bn128 import groth16_verify
    g = load_golden("verify.json")
    vk = g["verification_key"]
    assert len(g["cases"]) == 18 and sum(c["reference_verdict"] for c in g["cases"]) == 3
    for c in g["cases"]:
        assert groth16_verify(lib, vk, c["inputs"], c["proof"]) == c["reference_verdict"], (c["proof_file"], c["label"])


@pytest.mark.parametrize("name", ["t3", "t6"])
def test_golden_proofs_verify(lib, name):
    from wasmsnark_amd.bn128 import groth16_verify
    vk = json.load(open(os.path.join(GOLDEN, "keys", name + ".vk.json")))
    pub = json.load(open(os.path.join(GOLDEN, "keys", name + ".public.json")))
    for c in load_golden("proofs.json")[name]:
        assert c["reference_verifies"] and c["reference_rejects_wrong_public"]
        assert groth16_verify(lib, vk, pub, c["proof"]) is True
        wrong = [str((int(pub[0]) + 1) % R)] + pub[1:]
        assert groth16_verify(lib, vk, wrong, c["proof"]) is False
        swapped = dict(c["proof"], pi_a=c["proof"]["pi_c"])
        assert groth16_verify(lib, vk, pub, swapped) is False


def test_argument_errors(lib):
    from wasmsnark_amd import WsnarkError
    from wasmsnark_amd.bn128 import groth16_verify
    g = load_golden("verify.json")
    vk, c = g["verification_key"], g["cases"][0]
    with pytest.raises(ValueError):
        groth16_verify(lib, dict(vk, IC=vk["IC"][:3]), c["inputs"], c["proof"])           # fewer IC points than inputs + 1
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    bad = dict(c["proof"], pi_a=[str(q + 5), c["proof"]["pi_a"][1], "1"])
    with pytest.raises(WsnarkError):
        groth16_verify(lib, vk, c["inputs"], bad)                                          # coordinate not reduced mod q
    # a proof at infinity is well-formed and simply does not verify
    inf = {"pi_a": ["0", "1", "0"], "pi_b": [["0", "0"], ["1", "0"], ["0", "0"]], "pi_c": ["0", "1", "0"]}
    assert groth16_verify(lib, vk, c["inputs"], inf) is False
