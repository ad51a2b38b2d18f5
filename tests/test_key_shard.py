"""Points-sharded key handles (wsnark_pkey_load_shard): the reference's own worker split (src/bn128.js:353-361) with
GPUs as workers.  CPU emulator: every rank's handle holds 1/world of the points; the partial records of all ranks
combine (prove_finish) to the reference's own proof, for worlds that divide the key evenly, unevenly, and that leave
ranks without any pair."""
import json
import os

import pytest

from conftest import GOLDEN, load_golden
from emul_util import emul_bn128
from wasmsnark_amd import WsnarkError, formats


def _t(name):
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", name + ext), "rb").read()
    return rd(".pkey.bin"), rd(".witness.bin")


@pytest.mark.parametrize("name,world,table", [("t6", 2, "table"), ("t6", 8, "table"), ("t6", 70, "table"), ("t3", 4, "table"),
                                              ("t6", 3, "plain"), ("t6", 8, "plain")])
def test_point_shards_combine_to_the_reference_proof(tune, name, world, table):
    bn = emul_bn128()
    if table == "plain":
        tune(bn.lib, "KEY_TABLE", 0)
    pkey, wit = _t(name)
    sec = formats.pkey_bin_to_sections(pkey)
    whole = bn.load_key(sections=sec)
    assert whole.shard == {"rank": 0, "world": 1, "first_signal": 0, "n_signals": whole.n_vars, "n_hexps": whole.domain,
                           "h_interleave_log": 0}
    c = load_golden("proofs.json")[name][3]
    r, s = bytes.fromhex(c["r"]), bytes.fromhex(c["s"])
    recs, seen, seen_h, table_bytes = b"", 0, 0, 0
    for rank in range(world):
        key = bn.load_key(sections=sec, shard=(rank, world))
        sh = key.shard
        assert (sh["rank"], sh["world"]) == (rank, world) and sh["first_signal"] == seen
        seen += sh["n_signals"]
        seen_h += sh["n_hexps"]
        table_bytes += key.table["bytes"]
        if rank < world - 1:                                    # floor(n / world) each, the remainder to the last
            assert sh["n_signals"] == whole.n_vars // world and sh["n_hexps"] == whole.domain // world
        with pytest.raises(WsnarkError):
            bn.groth16GenProof(wit, key, r=r, s=s)              # a shard only yields partial sums
        with pytest.raises(WsnarkError):
            bn.groth16_prove_partial(wit, key, shard=((rank + 1) % world, world))   # ... and only its own
        recs += bn.groth16_prove_partial(wit, key, shard=(rank, world))
        key.free()
    assert seen == whole.n_vars and seen_h == whole.domain
    assert bn.groth16_prove_finish(whole, recs, r=r, s=s) == c["proof"]


def test_interleaved_hexps_share():
    """h_interleave_log = k: the handle's hExps share is the rank's rows of the 2^k-interleaved layout; the H sums of all
    ranks over the matching slices of h add up to the H sum of the whole key (checked through whole proofs: A, B1, C, B2
    from the records with WSNARK_PARTIAL_SKIP_H, H from wsnark_pkey_h_msm_dev on the interleaved slices)."""
    import ctypes as C
    bn = emul_bn128()
    pkey, wit = _t("t6")
    sec = formats.pkey_bin_to_sections(pkey)
    whole = bn.load_key(sections=sec)
    c = load_golden("proofs.json")["t6"][2]
    r, s = bytes.fromhex(c["r"]), bytes.fromhex(c["s"])
    h = bn.calcH(wit, sec["polsA"], sec["polsB"], whole.n_vars, whole.domain)           # plain form, natural order
    world, k = 4, 3
    m, cols = 1 << k, whole.domain >> k
    recs = b""
    for rank in range(world):
        key = bn.load_key(sections=sec, shard=(rank, world), h_interleave_log=k)
        assert key.shard["h_interleave_log"] == k and key.shard["n_hexps"] == whole.domain // world
        with pytest.raises(WsnarkError):
            bn.groth16_prove_partial(wit, key, shard=(rank, world))                    # no contiguous hExps range to sum h against
        rec = bn.groth16_prove_partial(wit, key, shard=(rank, world), skip_h=True)
        per = m // world
        mine = b"".join(h[32 * (rank * per + rr + m * j):32 * (rank * per + rr + m * j) + 32] for rr in range(per) for j in range(cols))
        buf = (C.c_uint8 * len(mine)).from_buffer_copy(mine)                            # (the emulator's "device" memory is host memory)
        hp = bn.h_multiexp_dev(key, C.addressof(buf), len(mine) // 32)
        with pytest.raises(WsnarkError):
            bn.h_multiexp_dev(key, C.addressof(buf), len(mine) // 32 - 1)
        recs += rec[:288] + hp + rec[384:]
        key.free()
    assert bn.groth16_prove_finish(whole, recs, r=r, s=s) == c["proof"]
    for bad in ((0, 4, 1), (0, 3, 3), (0, 4, 7)):                                      # 2^k < world; world not a power of two; 2^k > domain
        with pytest.raises(WsnarkError):
            bn.load_key(sections=sec, shard=bad[:2], h_interleave_log=bad[2])
    with pytest.raises(WsnarkError):
        bn.load_key(sections=sec, shard=(4, 4))
