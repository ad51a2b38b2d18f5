"""End-to-end prover checks that run without a GPU:
 * the oracle's groth16 prover reproduces the REFERENCE's proofs (tests/golden/proofs.json,
   produced by the reference prover with injected r, s and accepted by the reference verifier);
 * the toxic-waste closed form (wasmsnark_amd/synth.expected_proof) equals those proofs too,
   which is what validates it as the full-size end-to-end check of the GPU tests;
 * the product's kernel sources, run under the CPU thread emulator, give the same proofs."""
import json
import os

import pytest

from conftest import GOLDEN, load_golden, hamming_ok
from emul_util import emul_bn128
from gen_golden_keys import oracle_mul_base
from wasmsnark_amd import WsnarkError, synth

NAMES = ["t3", "t6"]


def _key(name):
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", name + ext), "rb").read()
    return rd(".pkey.bin"), rd(".witness.bin"), json.loads(rd(".meta.json"))


@pytest.mark.parametrize("name", NAMES)
def test_reference_accepted_its_own_proofs(name):
    for c in load_golden("proofs.json")[name]:
        assert c["reference_verifies"] is True
        assert c["reference_rejects_wrong_public"] is True


@pytest.mark.parametrize("name", NAMES)
def test_oracle_prover_matches_reference(orc, name):
    pkey, wit, _ = _key(name)
    for c in load_golden("proofs.json")[name]:
        got = orc.groth16_prove(wit, pkey, bytes.fromhex(c["r"]), bytes.fromhex(c["s"]), workers=8)
        assert got == c["proof"]
        got1 = orc.groth16_prove(wit, pkey, bytes.fromhex(c["r"]), bytes.fromhex(c["s"]), workers=1)
        assert got1 == c["proof"]


@pytest.mark.parametrize("name", NAMES)
def test_toxic_waste_closed_form_matches_reference(name):
    _, _, meta = _key(name)
    circ = synth.make_circuit(meta["log_domain"], n_public=meta["n_public"], seed=meta["circuit_seed"], style=meta.get("style", "rows"))
    S = synth.setup(circ, seed=meta["setup_seed"])
    for c in load_golden("proofs.json")[name]:
        exp = synth.expected_proof(circ, S, bytes.fromhex(c["r"]), bytes.fromhex(c["s"]), oracle_mul_base)
        assert exp == c["proof"]


@pytest.mark.parametrize("name", NAMES)
def test_emulated_kernels_prove_matches_reference(name):
    bn = emul_bn128()
    pkey, wit, _ = _key(name)
    key = bn.load_key(pkey)
    assert (key.n_vars, key.domain) == ((10, 8) if name == "t3" else (66, 64))
    for c in load_golden("proofs.json")[name]:
        got = bn.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"]))
        assert got == c["proof"]
    # random blinding: still a well-formed proof object (decimal strings, z = 1)
    p = bn.groth16GenProof(wit, pkey)
    assert p["pi_a"][2] == "1" and p["pi_b"][2] == ["1", "0"] and p["pi_c"][2] == "1"


@pytest.mark.parametrize("name", NAMES)
def test_emulated_prove_sparse_b_plan(tune, name):
    """Variables absent from matrix B have B1 = B2 = infinity; with enough of them the two B sums get their own
    plan that leaves those pairs out (forced here).  Same proofs, bit for bit."""
    bn = emul_bn128()
    tune(bn.lib, "PROVE_SPARSE", 2)
    pkey, wit, _ = _key(name)
    key = bn.load_key(pkey)
    for c in load_golden("proofs.json")[name]:
        assert bn.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("mode", ["plain", "h-only", "witness-only", "pieces", "pieces-sparse", "wide", "entry64", "slabs"])
def test_emulated_prove_key_table_modes(tune, name, mode):
    """Resident keys are fixed-base window tables by default (row w = 2^(c w) * section; one bucket set per sum).
    `plain` switches them off (the per-window path the MSM entry points use), `h-only` / `witness-only` keep tables for the hExps
    resp. for A / B1 / B2 / C alone (WSNARK_KEY_TABLE = 2 / 3: what the resident memory buys, section by section); `pieces`
    forces a table whose bucket set is cut into several tail pieces (the sum_v v T_v term of the host tail), also together with
    the masked plan variants; `wide` a window wider than the pair count suggests; `slabs` the row-per-launch build with a slab
    smaller than the sections (and three groups of rows behind an inversion each); `entry64` the 8-byte grouping entries a 2^24
    table key needs.  Same proofs, bit for bit."""
    bn = emul_bn128()
    lib = bn.lib
    if mode == "plain":
        tune(lib, "KEY_TABLE", 0)
    elif mode == "h-only":
        tune(lib, "KEY_TABLE", 2)
    elif mode == "witness-only":
        tune(lib, "KEY_TABLE", 3)
    elif mode.startswith("pieces"):
        tune(lib, "TABLE_C", 9)
        tune(lib, "TAIL_BITS", 6)
        if mode.endswith("sparse"):
            tune(lib, "PROVE_SPARSE", 2)
    elif mode == "wide":
        tune(lib, "TABLE_C", 13)
    elif mode == "slabs":
        tune(lib, "TABLE_C", 9)
        tune(lib, "TABLE_SLAB_LANES", 64)
    elif mode == "entry64":
        tune(lib, "MSM_ENTRY64", 1)
        tune(lib, "PROVE_SPARSE", 2)
    pkey, wit, _ = _key(name)
    key = bn.load_key(pkey)
    # wsnark_pkey_table_info: plain sections report one row; tables ceil(255 / c) rows of the window the mode asks for
    t = key.table
    assert (t["rows_w"], t["c_w"]) == ((1, 0) if mode in ("plain", "h-only") else (-(-255 // t["c_w"]), {"pieces": 9, "pieces-sparse": 9, "wide": 13, "slabs": 9}.get(mode, t["c_w"])))
    assert (t["rows_h"] == 1) == (mode in ("plain", "witness-only"))
    assert t["bytes"] == key.n_vars * 320 * t["rows_w"] + key.domain * 64 * t["rows_h"]
    for c in load_golden("proofs.json")[name]:
        assert bn.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]


@pytest.mark.parametrize("mode", ["table", "plain"])
def test_degenerate_key_points_against_the_oracle_prover(orc, tune, mode):
    """The prover is a pure function of (witness, key, r, s) -- the key's points need not come from a setup.  Here the
    point sections of the t6 key are made degenerate: runs of EQUAL points (the accumulation's doubling case: equal
    digits of equal points meet in one bucket), P / -P neighbours (the sum that passes through infinity) and points at
    infinity (x = 0), in A, B1, B2 and hExps.  The oracle's restatement of the reference prover is the judge, for the
    fixed-base table key and for the plain sections."""
    from primitives_common import degenerate_key_and_witness
    bn = emul_bn128()
    if mode == "plain":
        tune(bn.lib, "KEY_TABLE", 0)
    else:
        tune(bn.lib, "TABLE_SLAB_LANES", 64)
    key, w = degenerate_key_and_witness(orc, *_key("t6")[:2])
    k = bn.load_key(key)
    assert (k.table["rows_w"] > 1) == (mode != "plain")
    for r, s in ((bytes(32), bytes(32)), (bytes(range(1, 33)), bytes(range(101, 133)))):
        assert bn.groth16GenProof(w, k, r=r, s=s) == orc.groth16_prove(w, key, r, s, workers=8)


def test_proofs_before_the_table_rows_are_ready():
    """Round 4: a key load returns with the table rows still being built; proofs that arrive meanwhile plan on the plain sections
    (row 0 of every table-layout buffer), window-sharded partial calls wait for the rows (their sums must mean the same on every
    rank), and proofs after the build use the tables.  The emulator has no background: WSNARK_EMUL_TABLES_PENDING holds a key in the
    not-yet-ready state.  Same proofs in every state, bit for bit."""
    bn = emul_bn128()
    tune = bn.lib.tune
    try:
        for name in NAMES:
            pkey, wit, _ = _key(name)
            key = bn.load_key(pkey, wait_tables=False)
            assert key.table["rows_w"] > 1
            cases = load_golden("proofs.json")[name]
            tune("EMUL_TABLES_PENDING", 1)
            for j, c in enumerate(cases[:2]):
                r, s = bytes.fromhex(c["r"]), bytes.fromhex(c["s"])
                assert bn.groth16GenProof(wit, key, r=r, s=s) == c["proof"]                      # plain sections
                if j == 0 and name == NAMES[-1]:
                    recs = b"".join(bn.groth16_prove_partial(wit, key, shard=(g, 2)) for g in range(2))  # window shards: wait, tables
                    assert bn.groth16_prove_finish(key, recs, r=r, s=s) == c["proof"]
            tune("EMUL_TABLES_PENDING", None)
            key.wait_tables()
            c = cases[0]
            assert bn.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]       # tables
    finally:
        tune("EMUL_TABLES_PENDING", None)


def test_key_falls_back_to_plain_sections_when_a_table_does_not_fit(monkeypatch):
    """A table allocation the device refuses (emulator: WSNARK_EMUL_MAX_ALLOC) must not fail the load: the key keeps
    its plain sections (one row) and proves the same."""
    bn = emul_bn128()
    pkey, wit, _ = _key("t6")
    assert bn.load_key(pkey).table["rows_w"] > 1
    monkeypatch.setenv("WSNARK_EMUL_MAX_ALLOC", str(96 * 1024))      # B2 table: 66 x 128 B x 43 rows = 363 KB; plain 8 KB
    key = bn.load_key(pkey)
    assert key.table["rows_w"] == 1 and key.table["c_w"] == 0
    monkeypatch.delenv("WSNARK_EMUL_MAX_ALLOC")
    for c in load_golden("proofs.json")["t6"]:
        assert bn.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]


def test_emulated_mul_base_matches_oracle(orc):
    bn = emul_bn128()
    sc = b"".join(v.to_bytes(32, "little") for v in (0, 1, 2, 12345, orc.R - 1, orc.R, (1 << 256) - 1))
    assert bn.mul_base(1, sc) == oracle_mul_base(1, sc)
    assert bn.mul_base(2, sc) == oracle_mul_base(2, sc)


def test_key_format_errors():
    bn = emul_bn128()
    pkey, wit, _ = _key("t3")
    from wasmsnark_amd import WsnarkError
    with pytest.raises(WsnarkError):
        bn.load_key(pkey[:100])                       # truncated header
    bad = bytearray(pkey); bad[8:12] = (6).to_bytes(4, "little")   # domainSize not a power of two
    with pytest.raises(WsnarkError):
        bn.load_key(bytes(bad))
    bad = bytearray(pkey); bad[36:40] = (len(pkey)).to_bytes(4, "little")  # pHExps out of range
    with pytest.raises(WsnarkError):
        bn.load_key(bytes(bad))
    with pytest.raises(WsnarkError):
        bn.groth16GenProof(wit[:64], pkey)            # witness too short


def test_partial_finish_and_sections_loader():
    """prove_partial + prove_finish (world = 1 and a simulated world = 3) and the section-based key loader
    reproduce the reference's proofs."""
    import struct
    bn = emul_bn128()
    pkey, wit, _ = _key("t6")
    h = struct.unpack("<10I", pkey[:40])
    nv, npub, dom, pPA, pPB, pA, pB1, pB2, pC, pH = h
    sec = {"n_vars": nv, "n_public": npub, "domain": dom, "alfa1": pkey[40:104], "beta1": pkey[104:168],
           "delta1": pkey[168:232], "beta2": pkey[232:360], "delta2": pkey[360:488], "polsA": pkey[pPA:pPB],
           "polsB": pkey[pPB:pA], "pointsA": pkey[pA:pA + nv * 64], "pointsB1": pkey[pB1:pB1 + nv * 64],
           "pointsB2": pkey[pB2:pB2 + nv * 128], "pointsC": pkey[pC:pC + (nv - npub - 1) * 64],
           "pointsH": pkey[pH:pH + dom * 64]}
    key = bn.load_key(sections=sec)
    c = load_golden("proofs.json")["t6"][3]
    r, s = bytes.fromhex(c["r"]), bytes.fromhex(c["s"])
    assert bn.groth16GenProof(wit, key, r=r, s=s) == c["proof"]
    assert bn.groth16_prove_finish(key, bn.groth16_prove_partial(wit, key), r=r, s=s) == c["proof"]
    parts = b"".join(bn.groth16_prove_partial(wit, key, shard=(rank, 3)) for rank in range(3))
    assert bn.groth16_prove_finish(key, parts, r=r, s=s) == c["proof"]
    assert bn.last_blinding() == (r, s)
    # a section shorter than the header implies is refused, not read past
    for name in ("pointsA", "pointsB2", "pointsC", "pointsH"):
        short = dict(sec)
        short[name] = sec[name][:-1]
        with pytest.raises(WsnarkError):
            bn.load_key(sections=short)
    bad = dict(sec)
    bad["n_public"] = 0xFFFFFFFF                       # nPublic + 1 must not wrap around
    with pytest.raises(WsnarkError):
        bn.load_key(sections=bad)


def test_two_proofs_in_flight_on_one_key():
    """Lanes: several host threads prove with ONE key handle at once (each call takes a lane: its own queues, MSM plans,
    scratch, witness / h buffers); a third caller waits for a lane.  Every proof must equal the reference's."""
    import threading
    bn = emul_bn128()
    pkey, wit, _ = _key("t6")
    key = bn.load_key(pkey)
    cases = load_golden("proofs.json")["t6"]
    errors = []

    def worker(c):
        try:
            for _ in range(2):
                got = bn.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"]))
                if got != c["proof"]:
                    errors.append("mismatch")
                if bn.last_blinding() != (bytes.fromhex(c["r"]), bytes.fromhex(c["s"])):
                    errors.append("last_blinding is not per thread")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(c,)) for c in cases[:3]]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


def test_default_blinding_is_exposed_and_consistent(orc):
    """r, s left to the library (OS CSPRNG): the values it drew are readable like the reference's _pr / _ps
    (src/bn128.js:662-664) and reproduce the same proof when injected."""
    bn = emul_bn128()
    pkey, wit, _ = _key("t3")
    p1 = bn.groth16GenProof(wit, pkey)
    r1, s1 = bn.last_blinding()
    p2 = bn.groth16GenProof(wit, pkey)
    r2, s2 = bn.last_blinding()
    assert (r1, s1) != (r2, s2) and p1 != p2                       # fresh randomness per proof
    # the reference's own check of what it drew (test/bn128_prover.js:65-71): 96..160 zeros among the significant bits
    assert all(hamming_ok(v) for v in (r1, s1, r2, s2)), [v.hex() for v in (r1, s1, r2, s2)]
    assert bn.groth16GenProof(wit, pkey, r=r1, s=s1) == p1
    assert bn.groth16GenProof(wit, pkey, r=r2, s=s2) == p2
    assert orc.groth16_prove(wit, pkey, r1, s1) == p1              # and it is the proof the reference algorithm gives for them


def test_column_style_circuit_closed_form_and_oracle(orc):
    """The benchmark's circuit shape (1-3 non-zeros per COLUMN, every variable present):
    witness valid, every key point of A / B1 / B2 real (but the last variable's), and three provers agree:
    toxic-waste closed form == oracle (reference algorithm) == emulated kernels."""
    bn = emul_bn128()
    circ = synth.make_circuit(5, n_public=2, seed=123, style="columns")
    assert sum(1 for c in circ.A if not c) <= 1 and sum(1 for c in circ.B if not c) <= 1
    assert all(1 <= len(c) <= 4 for c in circ.B[:-1])
    S = synth.setup(circ, seed=9)
    pkey, _ = synth.build_key(circ, S, oracle_mul_base)
    wit = synth.witness_bin(circ)
    r, s = bytes(range(1, 33)), bytes(range(100, 132))
    want = synth.expected_proof(circ, S, r, s, oracle_mul_base)
    assert orc.groth16_prove(wit, pkey, r, s) == want
    assert bn.groth16GenProof(wit, pkey, r=r, s=s) == want


def test_staging_ring_and_chunked_witness_upload(orc):
    """Round 4: host buffers go through a pinned ring chunk by chunk (worker threads copy, the calling thread queues one DMA per
    chunk) and a proof from a HOST witness takes the grouping pass's digit histogram on every chunk as it lands.  The emulator
    has no DMA to overlap, but the bookkeeping is the product's: slots reused once their chunk has been handed on (20 chunks
    through a 16-slot ring), worker slices, chunk bounds clipped to the signals a handle sums, histogram pieces that add up."""
    import random
    bn = emul_bn128()
    tune = bn.lib.tune
    tune("STAGE_FORCE_RING", 1); tune("STAGE_RING_KB", 1024); tune("STAGE_CHUNK_KB", 64); tune("STAGE_WORKERS", 3)
    try:
        rnd = random.Random(77)
        n_el = 40960
        x = rnd.randbytes(31 * n_el)
        x = b"".join(x[31 * i:31 * i + 31] + b"\x00" for i in range(n_el))          # 1.25 MiB of reduced elements: 20 chunks through 16 slots
        assert bn.toMontgomeryN(x) == orc.to_mont_n(x)
        # a proof from a host witness of 2 chunks (2^11 constraints + change: the last chunk is ragged), whole key and two points shards
        circ = synth.NativeCircuit(bn.lib, 11, n_public=3, seed=21, style="columns")
        sec, _ = circ.build_sections()
        wit = circ.witness_bin()
        assert len(wit) > 65536 and len(wit) % 65536
        r, s = bytes(range(3, 35)), bytes(range(50, 82))
        want = circ.expected_proof(r, s)
        key = bn.load_key(sections=sec)
        assert bn.groth16GenProof(wit, key, r=r, s=s) == want
        key.free()
        recs = b""
        for rank in range(2):                                                        # shards: a chunk that straddles a shard's first / last signal
            k = bn.load_key(sections=sec, shard=(rank, 2))
            recs += bn.groth16_prove_partial(wit, k, shard=(rank, 2))
            if rank < 1:
                k.free()
        assert bn.groth16_prove_finish(k, recs, r=r, s=s) == want
    finally:
        for name in ("STAGE_FORCE_RING", "STAGE_RING_KB", "STAGE_CHUNK_KB", "STAGE_WORKERS"):
            tune(name, None)


def test_reduction_tail_geometries(orc):
    """The reduction tail's geometry is chosen by size (chunks of 4 / 8 buckets, pieces of 2^11 / 2^15 buckets whose rows are folded
    on the GPU by msm_rows; the G2 tail runs with the extension's components on lane pairs).  None of it may move a bit: stand-alone
    sums (per-window plans, windows cut into pieces) against the oracle, whole proofs (table plans: ONE bucket set cut into pieces)
    against the closed form, over chunk and piece sizes."""
    import itertools
    import random
    bn = emul_bn128()
    tune = bn.lib.tune
    names = ("MSM_CHUNK", "TAIL_BITS", "TAIL_L2")
    try:
        rnd = random.Random(5)
        for g, n in ((1, 1500), (2, 400)):
            ks = b"".join(rnd.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n))
            pts = bn.mul_base(g, ks)
            sc = b"".join((rnd.randrange(orc.R) if i % 7 else i % 3).to_bytes(32, "little") for i in range(n))
            want = orc.g_affine(g, orc.multiexp(g, "multiexp2" if g == 1 else "multiexp", sc, pts, n))
            msm = bn.g1_multiexp if g == 1 else bn.g2_multiexp
            # (l2: the second chunk level, msm_chunks2 -- m2 chunk pairs folded into one before the trees)
            for chunk, bits_w, l2 in ((2, 3, None), (4, None, None), (4, 4, None), (8, 3, None), (8, None, None), (2, 5, 2), (2, 6, 4), (2, 4, 2)):
                tune("MSM_CHUNK", chunk); tune("TAIL_BITS", bits_w); tune("TAIL_L2", l2)
                assert msm(sc, pts) == want, (g, chunk, bits_w, l2)
        for name in names:
            tune(name, None)
        circ = synth.NativeCircuit(bn.lib, 10, n_public=3, seed=8, style="columns")
        sec, _ = circ.build_sections()
        wit = circ.witness_bin()
        r, s = bytes(range(9, 41)), bytes(range(60, 92))
        want = circ.expected_proof(r, s)
        key = bn.load_key(sections=sec)
        assert key.table["rows_w"] > 1                      # table plans: one bucket set of 2^(c-1) buckets per sum
        for chunk, bits, l2 in ((2, 4, None), (4, None, None), (4, 6, None), (8, 4, None), (8, 6, None), (8, None, None), (2, None, 4), (2, 6, 4), (2, 6, 2), (4, 7, 2), (2, 7, 8)):
            tune("MSM_CHUNK", chunk); tune("TAIL_BITS", bits); tune("TAIL_L2", l2)
            assert bn.groth16GenProof(wit, key, r=r, s=s) == want, (chunk, bits, l2)
    finally:
        for name in names:
            tune(name, None)
