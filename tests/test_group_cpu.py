"""Several contexts in ONE process (csrc/group.hip): wsnark_group_* on the CPU thread emulator.  The emulator has one "device", so every
context of a group sits on it -- what is under test is the orchestration: one host thread per context, points shards of one key, the
distributed prover on every context at once with the in-library transport (device-to-device block copies between the contexts'
exchange buffers, ordered by host barriers; the records gathered in host memory), and the fall-back with CALC_H complete on every
device for group sizes the four-step transform cannot take.  Expected values: the REFERENCE's own proofs (tests/golden/proofs.json)."""
import json
import os

import pytest

import base64

from conftest import GOLDEN, hamming_ok, load_golden
from emul_util import emul_bn128
from wasmsnark_amd import bn128


def _key(name):
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", name + ext), "rb").read()
    return rd(".pkey.bin"), rd(".witness.bin"), json.loads(rd(".meta.json"))


@pytest.mark.parametrize("name,world", [("t6", 1), ("t6", 2), ("t6", 4), ("t6", 3), ("t3", 2), ("t3", 4)])
def test_group_prove_matches_reference(name, world):
    bn = emul_bn128()
    pkey, wit, _ = _key(name)
    g = bn128.Group(lib=bn.lib, devices=[0] * world)
    try:
        key = g.load_key(pkey)
        dom = 8 if name == "t3" else 64
        assert (key.world, key.domain) == (world, dom)
        # power-of-two groups of at most 2^floor(log2(domain) / 2) devices run CALC_H on the distributed transform
        log_n = dom.bit_length() - 1
        assert key.distributed_calc_h == (world & (world - 1) == 0 and (1 << (log_n // 2)) >= world)
        for c in load_golden("proofs.json")[name]:
            got = g.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"]))
            assert got == c["proof"]
        # drawn blinding: rank 0 draws, every rank assembles the same proof; it must be a well-formed proof object
        p = g.groth16GenProof(wit, key)
        assert p["pi_a"][2] == "1" and p["pi_b"][2] == ["1", "0"] and p["pi_c"][2] == "1"
        r_used, s_used = g.last_blinding()                      # (wsnark_group_last_blinding: the reference's _pr / _ps)
        assert hamming_ok(r_used) and hamming_ok(s_used)        # test/bn128_prover.js:65-71
        assert g.groth16GenProof(wit, key, r=r_used, s=s_used) == p
        key.free()
    finally:
        g.terminate()


def test_group_msm_matches_single_context():
    """wsnark_group_g{1,2}_msm: the reference's split of the pairs over the workers (src/bn128.js:353-415), partial sums added on the host"""
    bn = emul_bn128()
    golden = load_golden("msm.json")
    g = bn128.Group(lib=bn.lib, devices=[0, 0, 0])
    try:
        for which, name in ((0, "g1"), (1, "g2")):
            for c in golden[name]:
                if c["flavour"] == "accumulate_into_3G":
                    continue
                sc, pt = base64.b64decode(c["scalars"]), base64.b64decode(c["points"])
                many = g.g2_multiexp(sc, pt) if which else g.g1_multiexp(sc, pt)
                assert many == bytes.fromhex(c["multiexp_affine"]), (name, c["n"], c["flavour"])      # the reference's own result
    finally:
        g.terminate()


def test_group_errors_are_agreed_and_the_group_survives():
    bn = emul_bn128()
    pkey, wit, _ = _key("t6")
    g = bn128.Group(lib=bn.lib, devices=[0, 0])
    try:
        key = g.load_key(pkey)
        with pytest.raises(Exception):
            g.groth16GenProof(wit[:-32], key, r=bytes(32), s=bytes(32))       # witness too short: every rank must leave the collectives
        c = load_golden("proofs.json")["t6"][0]
        assert g.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]
        with pytest.raises(Exception):
            g.load_key(pkey[:200])
        key.free()
    finally:
        g.terminate()


def test_terminate_with_a_live_key_then_free_is_a_no_op():
    """wsnark_group_free deletes the group's keys; a GroupKey.free() / __del__ after Group.terminate() used to hand the
    dead handle to wsnark_group_pkey_free (use-after-free + double free).  The group now forgets its keys' handles when it dies."""
    bn = emul_bn128()
    pkey, wit, _ = _key("t3")
    g = bn128.Group(lib=bn.lib, devices=[0, 0])
    key, key2 = g.load_key(pkey), g.load_key(pkey)
    c = load_golden("proofs.json")["t3"][0]
    assert g.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]
    key2.free()                      # an explicit free while the group lives is the library's
    g.terminate()
    assert not key._h and not key2._h
    key.free(); key.free(); del key  # no-ops
    g.terminate()                    # idempotent
    g2 = bn128.Group(lib=bn.lib, devices=[0])          # the library is still healthy
    k2 = g2.load_key(pkey)
    assert g2.groth16GenProof(wit, k2, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]
    g2.terminate()


@pytest.mark.parametrize("devices", [[0, 1], [1, 0, 1, 0], [1, 1]])
def test_group_on_several_emulated_devices(monkeypatch, devices):
    """The emulator shows WSNARK_EMUL_DEVICES devices (tests/emul/hip_emul.h): a thread has a current device, queues and events belong to
    the device they were created on, and work on a queue of another device than the thread's current one -- or an event of one device
    recorded on a queue of another -- fails as the runtime makes it fail.  A group over two devices must therefore get every thread it
    uses onto the right device: the members' workers, the key load's helper thread (which round 6 found on the default context: on a
    second GPU its staging ring's events are device 0's), the caller's own thread in create / load / prove / free."""
    monkeypatch.setenv("WSNARK_EMUL_DEVICES", "2")
    bn = emul_bn128()
    pkey, wit, _ = _key("t6")
    g = bn128.Group(lib=bn.lib, devices=devices)
    try:
        key = g.load_key(pkey)
        assert key.world == len(devices)
        for c in load_golden("proofs.json")["t6"]:
            assert g.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]
        c = load_golden("msm.json")["g1"][3]
        sc, pt = base64.b64decode(c["scalars"]), base64.b64decode(c["points"])
        assert g.g1_multiexp(sc, pt) == bn.g1_multiexp(sc, pt)
        # the default context (device 0) still works from this thread afterwards: the group calls put the caller's device back
        k0 = bn.load_key(pkey)
        c0 = load_golden("proofs.json")["t6"][0]
        assert bn.groth16GenProof(wit, k0, r=bytes.fromhex(c0["r"]), s=bytes.fromhex(c0["s"])) == c0["proof"]
        k0.free()
        key.free()
    finally:
        g.terminate()
    with pytest.raises(Exception):
        bn128.Group(lib=bn.lib, devices=[0, 2])                 # no such device: refused, not wrapped
