"""BASELINE config 5 on the hardware the driver has (ONE MI355X): 2^22 and 2^24 constraints through the section
container (wsnark_pkey_load_sections: the key is past proving_key.bin's 4 GiB at 2^24), fixed-base tables on, the
8-byte grouping entries a 2^24 table key takes on its own, and the 8-way split of the north star run rank after rank on
the one GPU: window shards of the whole key, and the POINTS shards (wsnark_pkey_load_shard: 1/8 of the key resident per
rank, the reference's own worker split src/bn128.js:353-361) -- 8 records, prove_finish, equal to the one-call proof.

Expected values: the toxic-waste closed form of the library's host-side generator (csrc/synth.hip, pinned against the
oracle prover and the native verifier in tests/test_synth_native.py).  The group elements of the expectation (a*G1, b*G2,
c*G1) come from the product's own fixed-base helper `mul_base_kernel` -- a different code path from the prover (one lane
per scalar, double-and-add on the saturated 4x64 field), itself pinned by the C1 key hash
(test_gpu_parity.py::test_c1_example_witness_vs_reference_proof) and by smoke()'s oracle comparison."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bn():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__
    __graft_entry__.ensure_built()
    import wasmsnark_amd
    b = wasmsnark_amd.build(device=0)
    assert b.lib.path.endswith("wasmsnark_amd/libwsnark.so")
    return b


def _mem_ok(logd):
    import torch
    free_dev, _ = torch.cuda.mem_get_info(0)
    try:
        host_gb = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
    except (ValueError, OSError):
        host_gb = 0
    need_dev = {22: 40, 24: 140}[logd] * 2**30          # whole table key + one shard + per-lane scratch
    return free_dev >= need_dev and host_gb >= {22: 16, 24: 96}[logd]


@pytest.mark.parametrize("logd", [22, 24])
def test_config5_whole_key_window_shards_and_point_shards(bn, logd):
    from wasmsnark_amd import synth
    if logd == 24 and os.environ.get("WSNARK_TEST_2P24", "1") == "0":
        pytest.skip("WSNARK_TEST_2P24=0")
    if not _mem_ok(logd):
        pytest.skip("not enough free device / host memory for a 2^%d table key" % logd)
    world = 8
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=logd, style="columns")
    sec, _ = circ.build_sections()
    wit = circ.witness_bin()
    nv, dom = circ.n_vars, circ.domain
    assert dom == 1 << logd and nv == dom + 2
    key = bn.load_key(sections=sec)
    # the shape a key of this size takes on its own: tables, c = 20, 13 rows (and, at 2^24, 8-byte grouping entries:
    # 13 x 2^24 table indices + sign + 8 low bucket bits no longer fit 32 bits)
    assert key.table["c_w"] == 20 and key.table["rows_w"] == 13 and key.table["rows_h"] == 13
    assert key.table["bytes"] == 13 * (nv * 320 + dom * 64)
    r, s = os.urandom(32), os.urandom(32)
    want = circ.expected_proof(r, s)
    assert bn.groth16GenProof(wit, key, r=r, s=s) == want
    # (ii) the north star's wording: every rank sees all pairs and owns the table rows w = rank mod 8
    recs = b"".join(bn.groth16_prove_partial(wit, key, shard=(rank, world)) for rank in range(world))
    assert bn.groth16_prove_finish(key, recs, r=r, s=s) == want
    whole_bytes = key.table["bytes"]
    key.free()
    # (i) points shards: 1/8 of the key per rank, every row, uniform work
    recs, shares, sizes = b"", [], []
    for rank in range(world):
        k = bn.load_key(sections=sec, shard=(rank, world))
        sh = k.shard
        assert (sh["rank"], sh["world"], sh["first_signal"]) == (rank, world, rank * (nv // world))
        shares.append(sh["n_signals"] * k.table["rows_w"] + sh["n_hexps"] * k.table["rows_h"])      # (row, pair) work of the rank
        sizes.append(k.table["bytes"])
        assert k.table["rows_w"] > 1                                                                    # tables, not plain sections
        recs += bn.groth16_prove_partial(wit, k, shard=(rank, world))
        if rank < world - 1:
            k.free()
    assert bn.groth16_prove_finish(k, recs, r=r, s=s) == want          # (finish needs only the five fixed points: any handle)
    k.free()
    assert max(shares) <= 1.05 * min(shares), shares                   # balanced: within 5 %
    # about 1/8 of the memory each (a shard of n/8 pairs may take one more table row: 14 instead of 13 at 2^22)
    assert all(b <= whole_bytes / world * 1.10 for b in sizes), (sizes, whole_bytes)


def test_two_first_proofs_race_for_the_lazy_twiddle_tables():
    """the full per-pass twiddle tables of a transform plan are built lazily by the first caller, and two
    lanes used to race for them (host: both saw `not built`; device: the second lane read the table on its own queue
    while the builder's kernel was still queued).  A FRESH process (fresh context, no plan cached) fires two first proofs
    at 2^17 (three passes: tables needed) at the same instant, several times with new contexts."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for attempt in range(3):
        p = subprocess.run([sys.executable, os.path.join(here, "first_proofs_worker.py"), "17"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "first proofs OK" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
