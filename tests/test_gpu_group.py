"""-m gpu tests of SEVERAL CONTEXTS IN ONE PROCESS (wsnark_group_*, csrc/group.hip): what the Node drop-in uses to shard a proof over
the GPUs of a node without a second process.  A gpurun box has one GPU, so a group here is two (or four) contexts on device 0 -- the
real kernels, queues, device-to-device block copies and cross-context event waits; on a multi-GPU box the same test also runs over
distinct devices.  Expected values: the reference's golden proofs and the toxic-waste closed form."""
import json
import os

import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bn():
    import wasmsnark_amd
    return wasmsnark_amd.build(device=0)


def _devices(world):
    import torch
    n = torch.cuda.device_count()
    return [g % n for g in range(world)] if n > 1 else [0] * world


@pytest.mark.parametrize("world", [1, 2, 4, 3])
def test_group_proves_the_reference_proofs(bn, world):
    from wasmsnark_amd import bn128
    g = bn128.Group(lib=bn.lib, devices=_devices(world))
    try:
        for name in ("t3", "t6"):
            rd = lambda ext: open(os.path.join(GOLDEN, "keys", name + ext), "rb").read()
            pkey, wit = rd(".pkey.bin"), rd(".witness.bin")
            key = g.load_key(pkey)
            assert key.world == world
            for c in load_golden("proofs.json")[name]:
                assert g.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"], (name, world)
            key.free()
    finally:
        g.terminate()


@pytest.mark.parametrize("logd,world", [(13, 2), (18, 2), (18, 4), (20, 2), (16, 3)])
def test_group_proofs_at_size_equal_the_closed_form_and_the_single_gpu_proofs(bn, logd, world):
    """2^13 (odd log n: the transform's two splits alternate), 2^18, 2^20 over groups of two and four contexts, and a group of three
    (CALC_H complete on every device): the group's proof == the closed form == the one-GPU prover's, for injected and for drawn
    blinding values; the group's G1 / G2 sums over the same points == the single-context sums."""
    from wasmsnark_amd import bn128, synth
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=60 + logd)
    sec, _ = circ.build_sections()
    wit = circ.witness_bin()
    r, s = os.urandom(32), os.urandom(32)
    want = circ.expected_proof(r, s)
    g = bn128.Group(lib=bn.lib, devices=_devices(world))
    try:
        key = g.load_key(sections=sec)
        assert key.distributed_calc_h == (world & (world - 1) == 0)
        for _ in range(3):
            assert g.groth16GenProof(wit, key, r=r, s=s) == want
        one = bn.load_key(sections=sec)
        assert bn.groth16GenProof(wit, one, r=r, s=s) == want
        one.free()
        p = g.groth16GenProof(wit, key)                     # drawn blinding
        assert p["pi_a"][2] == "1" and p != want
        key.free()
        n = 5000
        sc = os.urandom(32 * n)
        pts1, pts2 = sec["pointsA"][:64 * n], sec["pointsB2"][:128 * n]
        assert g.g1_multiexp(sc, pts1) == bn.g1_multiexp(sc, pts1)
        assert g.g2_multiexp(sc, pts2) == bn.g2_multiexp(sc, pts2)
    finally:
        g.terminate()


def test_group_survives_a_failed_call_and_frees_what_is_left(bn):
    from wasmsnark_amd import bn128
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", "t6" + ext), "rb").read()
    pkey, wit = rd(".pkey.bin"), rd(".witness.bin")
    g = bn128.Group(lib=bn.lib, devices=_devices(2))
    key = g.load_key(pkey)
    with pytest.raises(Exception):
        g.groth16GenProof(wit[:-32], key, r=bytes(32), s=bytes(32))
    c = load_golden("proofs.json")["t6"][0]
    assert g.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]
    g.terminate()                                           # with the key still loaded: the group frees it ...
    assert not key._h                                       # ... and the wrapper forgets the handle, so that
    key.free()                                              # a late free() is a no-op (it was a use-after-free)
