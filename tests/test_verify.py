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
        return _lib.Lib()              # no init(): verification needs no GPU
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


# ---- malformed points: off-curve coordinates and twist points outside the order-r subgroup are INVALID ----
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def _f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def _f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % Q, Q - 2, Q)
    return (a[0] * n % Q, (-a[1]) * n % Q)


def _f2_sqrt(a):
    """sqrt in Fq[u]/(u^2 + 1), q = 3 mod 4 (complex method); None if a is not a square."""
    if a == (0, 0):
        return (0, 0)
    norm = (a[0] * a[0] + a[1] * a[1]) % Q
    s = pow(norm, (Q + 1) // 4, Q)
    if s * s % Q != norm:
        return None
    half = pow(2, Q - 2, Q)
    for sign in (1, -1):
        t = (a[0] + sign * s) * half % Q
        x0 = pow(t, (Q + 1) // 4, Q)
        if x0 * x0 % Q == t and x0:
            x1 = a[1] * pow(2 * x0 % Q, Q - 2, Q) % Q
            if _f2_mul((x0, x1), (x0, x1)) == (a[0] % Q, a[1] % Q):
                return (x0, x1)
    return None


def _twist_point_outside_g2():
    """A point on the twist y^2 = x^3 + 3/(9 + u) (src/bn128/build_bn128.js:79-90) -- almost surely NOT in the order-r
    subgroup (the twist's cofactor is ~2^254)."""
    b2 = _f2_mul((3, 0), _f2_inv((9, 1)))
    k = 1
    while True:
        x = (k, 7 * k + 1)
        x3 = _f2_mul(_f2_mul(x, x), x)
        y = _f2_sqrt(((x3[0] + b2[0]) % Q, (x3[1] + b2[1]) % Q))
        if y is not None:
            return x, y
        k += 1


def test_malformed_points_are_invalid_not_paired(lib):
    from wasmsnark_amd.bn128 import groth16_verify
    vk = json.load(open(os.path.join(GOLDEN, "keys", "t6.vk.json")))
    pub = json.load(open(os.path.join(GOLDEN, "keys", "t6.public.json")))
    good = load_golden("proofs.json")["t6"][1]["proof"]
    assert groth16_verify(lib, vk, pub, good) is True
    # G1 off the curve (y + 1), in pi_a, pi_c, the key's alfa1 and an IC point
    bump = lambda p: [p[0], str((int(p[1]) + 1) % Q), p[2]]
    assert groth16_verify(lib, vk, pub, dict(good, pi_a=bump(good["pi_a"]))) is False
    assert groth16_verify(lib, vk, pub, dict(good, pi_c=bump(good["pi_c"]))) is False
    assert groth16_verify(lib, dict(vk, vk_alfa_1=bump(vk["vk_alfa_1"])), pub, good) is False
    assert groth16_verify(lib, dict(vk, IC=[bump(vk["IC"][0])] + vk["IC"][1:]), pub, good) is False
    # G2 off the twist
    b = good["pi_b"]
    assert groth16_verify(lib, vk, pub, dict(good, pi_b=[b[0], [b[1][0], str((int(b[1][1]) + 1) % Q)], b[2]])) is False
    # G2 ON the twist but outside the order-r subgroup: in pi_b and as the key's gamma2 / delta2
    x, y = _twist_point_outside_g2()
    rogue = [[str(x[0]), str(x[1])], [str(y[0]), str(y[1])], ["1", "0"]]
    assert groth16_verify(lib, vk, pub, dict(good, pi_b=rogue)) is False
    assert groth16_verify(lib, dict(vk, vk_gamma_2=rogue), pub, good) is False
    assert groth16_verify(lib, dict(vk, vk_delta_2=rogue), pub, good) is False
    # the z coordinates of the proof are ignored like the reference does (src/bn128.js:741-760 force z = 1)
    assert groth16_verify(lib, vk, pub, dict(good, pi_a=[good["pi_a"][0], good["pi_a"][1], "0"], pi_c=good["pi_c"][:2] + ["5"])) is True


def test_input_count_cannot_wrap_the_size_check(lib):
    """(n_inputs + 1) * 64 wraps for n_inputs near 2^58: the size comparison is done without the product."""
    import ctypes as C
    vkb = bytes(512)
    valid = C.c_int(7)
    for n_inputs in (1 << 58, (1 << 58) - 1, (1 << 64) - 1, 2):
        rc = lib.c.wsnark_groth16_verify(vkb, len(vkb), bytes(64), C.c_uint64(n_inputs), bytes(384), C.byref(valid))
        assert rc == 1 and valid.value == 0, n_inputs      # WSNARK_ERR_SIZE, nothing read past the key
