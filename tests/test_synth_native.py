"""The library's host-side circuit generator (csrc/synth.hip, wsnark_synth_*) is what the full-size GPU tests and
bench.py prove on, and its closed form is what they compare with -- so it is pinned here, without a GPU and without the
product's prover: the ORACLE's groth16 prover (pinned to the reference's own proofs in test_prove_cpu.py) must produce,
on a key this generator built, exactly the proof the generator's closed form predicts, and the native verifier must
accept it with the right public signals and reject it with a wrong one."""
import pytest

from emul_util import emul_bn128
from wasmsnark_amd import synth


@pytest.mark.parametrize("log_domain,style", [(4, "columns"), (6, "columns"), (6, "rows"), (9, "columns"), (6, "boolean"), (9, "boolean")])
def test_native_generator_against_oracle_prover_and_verifier(orc, log_domain, style):
    bn = emul_bn128()                                  # (only for mul_base and the host-side generator / verifier)
    circ = synth.NativeCircuit(bn.lib, log_domain, n_public=3, seed=11 + log_domain, style=style)
    assert circ.domain == 1 << log_domain and (circ.n_vars == circ.domain + 2 or style == "boolean")
    pkey, vk = circ.build_key()
    wit = circ.witness_bin()
    assert wit[:32] == (1).to_bytes(32, "little") and len(wit) == circ.n_vars * 32
    for r, s in ((bytes(32), bytes(32)), (b"\x07" + bytes(31), b"\x09" + bytes(31)), (b"\xff" * 32, b"\xfe" + b"\xff" * 31)):
        got = orc.groth16_prove(wit, pkey, r, s, workers=2)
        assert got == circ.expected_proof(r, s)
        pub = circ.public_signals()
        assert bn.groth16Verify(vk, pub, got)
        assert not bn.groth16Verify(vk, [str(int(pub[0]) + 1)] + pub[1:], got)


def test_native_generator_shapes():
    bn = emul_bn128()
    c = synth.NativeCircuit(bn.lib, 12, n_public=5, seed=1, style="columns")
    assert c.absent == (1, 1)                          # only the very last variable can occur nowhere
    assert 2 * c.n_vars <= c.nnz <= 6 * c.n_vars + 2 * c.domain
    r = synth.NativeCircuit(bn.lib, 12, n_public=5, seed=1, style="rows")
    assert r.absent[0] > c.n_vars // 4 and r.absent[1] > c.n_vars // 4
    # round 6, "boolean": bit decompositions -- groups of 14 free bits with their booleanity rows b (b - 1) = 0 (C = 0), a recomposition
    # row and a product row: >= 80 % of the witness is 0 / 1, the rest 14-bit values and full-size field elements
    b = synth.NativeCircuit(bn.lib, 12, n_public=5, seed=1, style="boolean")
    wb = b.witness_bin()
    vals = [int.from_bytes(wb[32 * i:32 * i + 32], "little") for i in range(b.n_vars)]
    assert sum(1 for v in vals if v in (0, 1)) >= 0.8 * b.n_vars
    assert sum(1 for v in vals if v >= 1 << 200) >= 0.04 * b.n_vars and b.n_vars <= b.domain + 2
    assert b.absent[1] >= b.n_vars // 20               # the recomposed values never occur in B: their B1 / B2 points are infinity
    sec, _ = c.build_sections()
    assert len(sec["pointsA"]) == c.n_vars * 64 and len(sec["pointsB2"]) == c.n_vars * 128
    assert len(sec["pointsC"]) == (c.n_vars - 6) * 64 and len(sec["pointsH"]) == c.domain * 64
    # the same seeds give the same circuit; another seed another one
    assert synth.NativeCircuit(bn.lib, 12, n_public=5, seed=1).witness_bin() == c.witness_bin()
    assert synth.NativeCircuit(bn.lib, 12, n_public=5, seed=2).witness_bin() != c.witness_bin()
