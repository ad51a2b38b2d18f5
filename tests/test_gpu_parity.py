"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
libwsnark.so, against (1) the reference-generated golden vectors, (2) the pinned CPU oracle on
seeded inputs at sizes it finishes in seconds, and (3) size-independent exact properties at
BASELINE.json's full sizes (NTT round trip / coset identity at 2^22 and 2^20, MSM of 2^20 pairs
against its closed form in the exponent, Groth16 proofs against the toxic-waste closed form).
Everything is integer arithmetic: the bar is bit-exact equality."""
import base64
import ctypes
import json
import os
import random

import pytest

from conftest import GOLDEN, load_golden
from wasmsnark_amd._lib import WsnarkError

pytestmark = pytest.mark.gpu
H = bytes.fromhex
B64 = base64.b64decode


@pytest.fixture(scope="module")
def bn():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import __graft_entry__
    __graft_entry__.ensure_built()     # hipcc is in the image: build in-tree if the snapshot came without the .so
    import wasmsnark_amd
    b = wasmsnark_amd.build(device=0)
    assert b.lib.path.endswith("wasmsnark_amd/libwsnark.so")   # the native HIP library, in-tree
    return b


def rand_fr(rnd, n):
    return b"".join(rnd.randrange(1 << 256).to_bytes(32, "little") for _ in range(n))


# ------------------------------------------------------------------ golden vectors
def test_fft_golden(bn):
    for c in load_golden("fft.json")["cases"]:
        x = B64(c["input_mont"])
        assert bn.fft(x, 0) == B64(c["fft0"])
        assert bn.fft(x, 1) == B64(c["fft1"])
        if c["n"] == 1:
            with pytest.raises(Exception):
                bn.ifft(x, 0)
        else:
            assert bn.ifft(x, 0) == B64(c["ifft0"])
            assert bn.ifft(x, 1) == B64(c["ifft1"])
    for n in (0, 3, 6, 1000):                     # the reference traps (src/build_fft.js:137-154)
        with pytest.raises(Exception):
            bn.fft(b"\0" * (32 * n), 0)


@pytest.mark.parametrize("g", [1, 2])
def test_msm_golden(bn, g):
    for c in load_golden("msm.json")["g%d" % g]:
        if c["flavour"] == "accumulate_into_3G":
            continue
        s, p = B64(c["scalars"]), B64(c["points"])
        out = bn.g1_multiexp(s, p) if g == 1 else bn.g2_multiexp(s, p)
        assert out == H(c["multiexp_affine"]), (g, c["n"], c["flavour"])


def test_calc_h_golden(bn):
    for c in load_golden("calch.json"):
        h = bn.calcH(B64(c["signals"]), B64(c["polsA"]), B64(c["polsB"]), c["nSignals"], c["domain"])
        assert h == B64(c["h"])


@pytest.mark.parametrize("name", ["t3", "t6"])
def test_proofs_golden(bn, name):
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", name + ext), "rb").read()
    key = bn.load_key(rd(".pkey.bin"))
    for c in load_golden("proofs.json")[name]:
        assert bn.groth16GenProof(rd(".witness.bin"), key, r=H(c["r"]), s=H(c["s"])) == c["proof"]


# ------------------------------------------------------------------ vs the oracle
@pytest.mark.parametrize("bits", [1, 4, 9, 10, 11, 14, 16, 17, 18])
def test_ntt_vs_oracle(bn, orc, bits):
    n = 1 << bits
    x = orc.to_mont_n(rand_fr(random.Random(bits), n))
    for odd in (0, 1):
        assert bn.fft(x, odd) == orc.fft(x, n, odd)
        assert bn.ifft(x, odd) == orc.fft(x, n, odd, inverse=True)


@pytest.mark.parametrize("bits", [5, 10, 13, 17])
def test_ntt_extreme_values_vs_oracle(bn, orc, bits):
    """Vectors of the largest residues (r - 1 everywhere, r - 1 / 0 / 1 patterns): the butterflies keep sums
    uncorrected in [0, 4p) and fold them back from [0, 16p), so these inputs drive every lazy bound to its limit."""
    n = 1 << bits
    top = (orc.R - 1).to_bytes(32, "little")
    pats = [top * n,
            b"".join(top if (i & 1) else (0).to_bytes(32, "little") for i in range(n)),
            b"".join(top if i < n // 2 else (1).to_bytes(32, "little") for i in range(n))]
    for x in pats:
        for odd in (0, 1):
            assert bn.fft(x, odd) == orc.fft(x, n, odd), (bits, odd)
            assert bn.ifft(x, odd) == orc.fft(x, n, odd, inverse=True), (bits, odd)


def test_montgomery_maps_vs_oracle(bn, orc):
    x = rand_fr(random.Random(9), 5000)
    xr = b"".join((int.from_bytes(x[i:i + 32], "little") % orc.R).to_bytes(32, "little") for i in range(0, len(x), 32))
    assert bn.toMontgomeryN(xr) == orc.to_mont_n(xr)
    assert bn.fromMontgomeryN(xr) == orc.from_mont_n(xr)


def _skewed_scalars(rnd, n, R):
    out = []
    for _ in range(n):
        u = rnd.random()          # example-witness histogram of SURVEY.md section 8d
        if u < 0.067: v = 0
        elif u < 0.098: v = 1
        elif u < 0.2: v = rnd.randrange(1 << 32)
        elif u < 0.22: v = (1 << 256) - 1 - rnd.randrange(1 << 30)      # >= r, raw 256-bit
        else: v = rnd.randrange(R)
        out.append(v.to_bytes(32, "little"))
    return b"".join(out)


@pytest.mark.parametrize("g,n", [(1, 1), (1, 63), (1, 4096), (1, 20000), (2, 1), (2, 1500)])
def test_msm_vs_oracle(bn, orc, g, n):
    rnd = random.Random(1000 * g + n)
    sz = 64 if g == 1 else 128
    pts = bytearray(bn.mul_base(g, b"".join(rnd.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n))))
    if n > 10:   # planted: infinity points, a duplicated point, a P / -P pair
        pts[3 * sz:4 * sz] = b"\0" * sz
        pts[5 * sz:6 * sz] = pts[4 * sz:5 * sz]
    sc = _skewed_scalars(rnd, n, orc.R)
    got = bn.g1_multiexp(sc, bytes(pts)) if g == 1 else bn.g2_multiexp(sc, bytes(pts))
    want = orc.g_affine(g, orc.multiexp(g, "workers8", sc, bytes(pts), n))
    assert got == want


@pytest.mark.parametrize("g", [1, 2])
def test_msm_corner_case_buckets(bn, orc, g):
    """the accumulation's fast loop hands every corner case of the reference's addition to the generic loop"""
    from primitives_common import check_msm_corner_case_buckets
    check_msm_corner_case_buckets(bn, orc, g, cases=40)


def test_msm_all_same_point(bn, orc):
    n = 3000
    pts = bn.mul_base(1, (7).to_bytes(32, "little")) * n
    sc = (5).to_bytes(32, "little") * n
    want = bn.mul_base(1, (35 * n % orc.R).to_bytes(32, "little"))
    assert bn.g1_multiexp(sc, pts)[:64] == want


@pytest.mark.parametrize("cfg", [{"MSM_ENTRY64": 1}, {"MSM_HOT_MIN": 2}, {"MSM_LMAX": 6}, {"MSM_C": 12, "MSM_HOT_MIN": 2, "MSM_LMAX": 4}])
def test_msm_grouping_variants_agree(bn, tune, cfg):
    """Entry width, hot-bucket threshold, task cap and window width only change how the pairs are grouped and how the partial sums of
    split buckets are folded (the three roles of the one combine launch): the sum must not move by a bit (G1 and G2, 60 000 pairs
    with 30 % ones)."""
    import numpy as np
    n = 60000
    rng = np.random.default_rng(42)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ones = rng.random(n) < 0.3
    sc[ones] = 0
    sc[ones, 0] = 1
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    for g in (1, 2):
        pts = bn.mul_base(g, ks.tobytes())
        msm = bn.g1_multiexp if g == 1 else bn.g2_multiexp
        base = msm(sc.tobytes(), pts)
        for k, v in cfg.items():
            tune(bn.lib, k, v)
        assert msm(sc.tobytes(), pts) == base
        assert msm(sc.tobytes(), pts) == base          # (twice: the hot-bucket completion counters are left at zero)
        for k in cfg:
            bn.lib.tune(k, None)


def test_msm_window_override(bn, orc, tune):
    rnd = random.Random(77)
    n = 2000
    pts = bn.mul_base(1, b"".join(rnd.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n)))
    sc = _skewed_scalars(rnd, n, orc.R)
    want = orc.g_affine(1, orc.multiexp(1, "workers8", sc, pts, n))
    for c in (4, 9, 13, 16):
        tune(bn.lib, "MSM_C", c)
        assert bn.g1_multiexp(sc, pts) == want, c


def test_calc_h_vs_oracle(bn, orc):
    import struct
    rnd = random.Random(31)
    nS, dom = 3000, 4096
    sig = b"".join(rnd.randrange(orc.R).to_bytes(32, "little") for _ in range(nS))
    def pols():
        out = bytearray()
        for s in range(nS):
            k = rnd.randrange(0, 4)
            out += struct.pack("<I", k)
            for idx in rnd.sample(range(dom), k):
                out += struct.pack("<I", idx) + rnd.randrange(orc.R).to_bytes(32, "little")
        return bytes(out)
    A, B = pols(), pols()
    assert bn.calcH(sig, A, B, nS, dom) == orc.calc_h(sig, A, B, nS, dom)


# ------------------------------------------------------------------ full-size properties
def _torch_bytes(t):
    return t.cpu().numpy().tobytes()


def test_ntt_2p22_roundtrip_and_coset_identity(bn):
    # reference test/fft.js:16-121 at BASELINE config 3 size, device-resident.
    # The _dev entry points run on the library's own stream: synchronise around them.
    import torch
    sync = torch.cuda.synchronize
    n = 1 << 22
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, generator=g)
    x[31::32] &= 0x1F                              # < 2^253 < r: valid field elements
    d = x.cuda()
    y = d.clone()
    sync()
    bn.fft_dev(y.data_ptr(), n, 0)
    sync()
    assert not torch.equal(y, d)
    bn.fft_dev(y.data_ptr(), n, 0, inverse=True)
    sync()
    assert torch.equal(y, d)                       # ifft(fft(x)) == x, bit for bit
    # coset identity at n = 2^21 -> 2n = 2^22: fft(x,0) and fft(x,1) interleaved == fft(pad(x), 2n)
    m = n // 2
    e, o = d[: m * 32].clone(), d[: m * 32].clone()
    big = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    big[: m * 32] = d[: m * 32]
    sync()
    bn.fft_dev(e.data_ptr(), m, 0)
    bn.fft_dev(o.data_ptr(), m, 1)
    bn.fft_dev(big.data_ptr(), n, 0)
    sync()
    inter = torch.stack([e.view(m, 32), o.view(m, 32)], dim=1).reshape(-1)
    assert torch.equal(inter, big)


def test_msm_2p20_closed_form(bn, orc):
    # BASELINE config 2: 2^20 pairs.  P_i = k_i G, so MSM == (sum s_i k_i mod r) G exactly.
    import torch
    n = 1 << 20
    rnd = random.Random(2020)
    ks = [rnd.randrange(1, orc.R) for _ in range(n)]
    pts = bn.mul_base(1, b"".join(k.to_bytes(32, "little") for k in ks))
    for flavour in ("uniform", "circuit"):
        if flavour == "uniform":
            ss = [rnd.randrange(orc.R) for _ in range(n)]
            sc = b"".join(s.to_bytes(32, "little") for s in ss)
        else:
            sc = _skewed_scalars(rnd, n, orc.R)
            ss = [int.from_bytes(sc[i:i + 32], "little") for i in range(0, n * 32, 32)]
        expect = sum(s * k for s, k in zip(ss, ks)) % orc.R
        want = bn.mul_base(1, expect.to_bytes(32, "little"))
        d_s = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
        d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        got = bn.g1_multiexp_dev(d_s.data_ptr(), d_p.data_ptr(), n)
        assert got[:64] == want, flavour
        assert got == bn.g1_multiexp(sc, pts)      # host-pointer boundary gives the same


@pytest.mark.parametrize("g", [1, 2])
def test_msm_full_size_adversarial_closed_forms(bn, orc, g):
    """2^20 (G1) / 2^18 (G2) pairs with inputs that stress the grouping and the hot-bucket paths: every scalar
    equal (16 buckets hold everything), cancelling +P / -P pairs (sum is infinity), scalars >= r up to
    2^256 - 1 (reduced mod r like the reference's double-and-add), all-zero scalars, x == 0 points."""
    import time
    import torch
    n = 1 << (20 if g == 1 else 18)
    esz = 64 if g == 1 else 128
    rnd = random.Random(77 + g)
    ks = [rnd.randrange(1, orc.R) for _ in range(n)]
    pts = bn.mul_base(g, b"".join(k.to_bytes(32, "little") for k in ks))
    d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    msm = bn.g1_multiexp_dev if g == 1 else bn.g2_multiexp_dev
    one = ((1 << 256) % orc.Q).to_bytes(32, "little")          # 1 in Montgomery form
    inf = bytes(esz // 2) + one + bytes(esz // 2 - 32)          # affine part of the reference's infinity (0, 1, 0)

    def run(sc_bytes, d_points=d_p):
        d_s = torch.frombuffer(bytearray(sc_bytes), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = msm(d_s.data_ptr(), d_points.data_ptr(), n)
        return out, time.perf_counter() - t0

    def expect(e):
        return bn.mul_base(g, (e % orc.R).to_bytes(32, "little"))

    ksum = sum(ks) % orc.R
    # (a) every scalar equal: one bucket per window holds all n entries
    s0 = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % orc.R
    out, dt = run(s0.to_bytes(32, "little") * n)
    assert out[:esz] == expect(s0 * ksum)
    assert dt < 2.0                                     # degraded (hot buckets), not pathological
    # (b) all scalars 2^256 - 1 (>= r: raw 256-bit values are legal, src/build_multiexp.js:651-744)
    big = (1 << 256) - 1
    out, _ = run(big.to_bytes(32, "little") * n)
    assert out[:esz] == expect(big * ksum)
    # (c) all-zero scalars: infinity = (0, 1, 0)
    out, _ = run(bytes(32 * n))
    assert out[:esz] == inf and out[esz:] == bytes(len(out) - esz)
    # (d) second half = negated first half, same scalars: everything cancels
    half = n // 2
    q = orc.Q
    neg = bytearray(pts[:half * esz])
    nested = neg
    for i in range(half):
        o = i * esz + esz // 2
        if g == 1:
            y = int.from_bytes(nested[o:o + 32], "little")
            nested[o:o + 32] = ((q - y) % q).to_bytes(32, "little")
        else:
            for c in (0, 32):
                y = int.from_bytes(nested[o + c:o + c + 32], "little")
                nested[o + c:o + c + 32] = ((q - y) % q).to_bytes(32, "little")
    d_pn = torch.frombuffer(bytearray(pts[:half * esz]) + nested, dtype=torch.uint8).cuda()
    sc_half = rand_fr(rnd, half)
    out, _ = run(sc_half + sc_half, d_pn)
    assert out[:esz] == inf
    # (e) x == 0 marks infinity whatever y is: zero the x of the odd points, they must drop out
    holes = bytearray(pts)
    xz = esz // 2
    for i in range(1, n, 2):
        holes[i * esz:i * esz + xz] = bytes(xz)
    d_ph = torch.frombuffer(holes, dtype=torch.uint8).cuda()
    ss = [rnd.randrange(orc.R) for _ in range(n)]
    out, _ = run(b"".join(v.to_bytes(32, "little") for v in ss), d_ph)
    assert out[:esz] == expect(sum(ss[i] * ks[i] for i in range(0, n, 2)))


@pytest.mark.parametrize("logd,style", [(12, "boolean"), (16, "boolean"), (20, "boolean")])
def test_prove_boolean_heavy_valid_circuit_vs_closed_form(bn, logd, style):
    """A VALID boolean-heavy circuit at full size (round 6; csrc/synth.hip style 2: bit decompositions with their booleanity rows
    b (b - 1) = 0, 87.5 % of the witness 0 / 1 -- what circom circuits look like): every window row's digit-1 bucket takes ~45 % of
    the pairs (very hot buckets: msm_plan_emit_hot, the hot role of msm_combine_all), the zero scalars drop out, one B column has an
    entry in every booleanity row.  Expected value: the toxic-waste closed form of the library's generator, which
    tests/test_synth_native.py pins against the ORACLE's prover and the verifier on the CPU."""
    from wasmsnark_amd import synth
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=logd, style=style)
    sec, _ = circ.build_sections()
    wit = circ.witness_bin()
    vals01 = sum(1 for i in range(0, len(wit), 32) if wit[i + 1:i + 32] == bytes(31) and wit[i] < 2)
    assert vals01 >= 0.8 * circ.n_vars
    key = bn.load_key(sections=sec)
    for r, s in ((bytes(32), bytes(32)), (bytes(range(32)), bytes(range(32, 64)))):
        assert bn.groth16GenProof(wit, key, r=r, s=s) == circ.expected_proof(r, s)
    # ... and before the table rows exist (plain sections: the per-window plans, whose digit-0 / digit-1 buckets are hot in EVERY window)
    k2 = bn.load_key(sections=sec, wait_tables=False)
    assert bn.groth16GenProof(wit, k2, r=bytes(32), s=bytes(32)) == circ.expected_proof(bytes(32), bytes(32))
    k2.free(); key.free(); circ.free()


@pytest.mark.parametrize("logd,style", [(10, "columns"), (16, "columns"), (16, "rows"), (20, "columns")])
def test_prove_vs_toxic_waste_closed_form(bn, logd, style):
    """(20, columns) is BASELINE config 4 at full size: the circuit SURVEY.md section 8d C4 specifies (1-3 non-zeros per
    column, every variable present).  'rows' leaves ~40 % of the A / B key points at infinity: the plan-variant path.
    The expectation is the toxic-waste closed form a*G1, b*G2, c*G1; those three group elements come from the product's own
    fixed-base helper (`mul_base_kernel`: one lane per scalar, double-and-add on the saturated 4x64 field -- not the prover's
    code path), which is itself pinned by the C1 key hash (test_c1_example_witness_vs_reference_proof) and by smoke()'s
    comparison with the oracle; the scalars a, b, c are plain Python integers (wasmsnark_amd/synth.py)."""
    from wasmsnark_amd import synth
    circ = synth.make_circuit(logd, n_public=5, seed=logd, style=style)
    S = synth.setup(circ, seed=3)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    key = bn.load_key(pkey)
    wit = synth.witness_bin(circ)
    for r, s in ((b"\0" * 32, b"\0" * 32), (b"\xff" * 32, b"\xfe" + b"\xff" * 31), (os.urandom(32), os.urandom(32))):
        got = bn.groth16GenProof(wit, key, r=r, s=s)
        assert got == synth.expected_proof(circ, S, r, s, bn.mul_base)
    # blinding left to the library (the reference's default path, src/bn128.js:642-661): read back what it drew
    # (its _pr / _ps, :662-664) and check the proof against the closed form for exactly those values
    got = bn.groth16GenProof(wit, key)
    r, s = bn.last_blinding()
    assert got == synth.expected_proof(circ, S, r, s, bn.mul_base)
    got2 = bn.groth16GenProof(wit, key)
    assert bn.last_blinding() != (r, s) and got2 != got
    from conftest import hamming_ok                      # the reference's own check of its draw: test/bn128_prover.js:65-71
    assert all(hamming_ok(v) for v in (r, s) + bn.last_blinding())
    key.free()


def test_host_witness_paths_and_runtime_switches(bn):
    """A witness in HOST memory reaches the GPU by two routes -- pageable memory through the pinned ring (chunk by chunk, the grouping
    histogram taken per arriving chunk; ring slots reused: 64 KiB chunks through a ring the first upload sized) and a PINNED buffer
    (wsnark_host_alloc) DMA'd in place -- and the reduction tail has its geometries behind run-time switches (wsnark_tuning_set).  Every combination must give
    the closed-form proof at 2^18 (129 chunks of the witness at the smallest chunk size: one more than the ring has slots), on the whole key and on a points shard."""
    import ctypes as C
    from wasmsnark_amd import synth
    circ = synth.NativeCircuit(bn.lib, 18, n_public=5, seed=44)
    sec, _ = circ.build_sections()
    wit = circ.witness_bin()
    r, s = os.urandom(32), os.urandom(32)
    want = circ.expected_proof(r, s)
    key = bn.load_key(sections=sec)
    pinned = C.c_void_p()
    bn.lib.check(bn.lib.c.wsnark_host_alloc(len(wit), C.byref(pinned)))
    names = ("STAGE_CHUNK_KB", "STAGE_WORKERS", "MSM_CHUNK", "TAIL_BITS", "TAIL_L2")
    try:
        C.memmove(pinned, wit, len(wit))
        # (TAIL_L2: the second chunk level of the tail, msm_chunks2 -- default 4 at this size; off, 2 and 8, and on other chunk / piece sizes)
        for cfg in ({}, {"STAGE_CHUNK_KB": 64, "STAGE_WORKERS": 3}, {"STAGE_CHUNK_KB": 1024}, {"MSM_CHUNK": 8, "TAIL_BITS": 15}, {"MSM_CHUNK": 2, "TAIL_BITS": 10},
                    {"TAIL_L2": 1}, {"TAIL_L2": 2}, {"TAIL_L2": 8}, {"TAIL_L2": 8, "MSM_CHUNK": 2, "TAIL_BITS": 9}, {"TAIL_L2": 2, "MSM_CHUNK": 16, "TAIL_BITS": 13}):
            for n in names:
                bn.lib.tune(n, cfg.get(n))
            assert bn.groth16GenProof(wit, key, r=r, s=s) == want, cfg
            assert bn.groth16GenProof_hostptr(pinned.value, len(wit), key, r=r, s=s) == want, cfg
        for n in names:
            bn.lib.tune(n, None)
        key.free()
        recs = b""
        for rank in range(2):
            k = bn.load_key(sections=sec, shard=(rank, 2))
            recs += bn.groth16_prove_partial(wit, k, shard=(rank, 2))
            if rank == 0:
                k.free()
        assert bn.groth16_prove_finish(k, recs, r=r, s=s) == want
        k.free()
    finally:
        for n in names:
            bn.lib.tune(n, None)
        bn.lib.c.wsnark_host_free(pinned)


def test_proofs_beside_the_background_table_build(bn):
    """Round 4: wsnark_pkey_load returns when the sections are resident; the rows of the fixed-base tables are built behind it on a
    queue of their own, and proofs that arrive meanwhile run on the plain sections.  At 2^18: proofs right behind the load (two of
    them from the two lanes at once), across the moment the tables become ready, and after wsnark_pkey_wait_tables -- all equal to
    the closed form; the same with a scratch slab smaller than the sections; a
    key freed while its build is still running; load stats report the build only once it is over."""
    import threading
    from wasmsnark_amd import synth
    circ = synth.NativeCircuit(bn.lib, 18, n_public=5, seed=45)
    sec, _ = circ.build_sections()
    wit = circ.witness_bin()
    r, s = os.urandom(32), os.urandom(32)
    want = circ.expected_proof(r, s)
    try:
        for cfg in ({}, {"TABLE_SLAB_LANES": 4096}):
            for n in ("TABLE_SLAB_LANES",):
                bn.lib.tune(n, cfg.get(n))
            key = bn.load_key(sections=sec, wait_tables=False)
            assert key.table["rows_w"] > 1
            got = [None, None]
            th = [threading.Thread(target=lambda i=i: got.__setitem__(i, bn.groth16GenProof(wit, key, r=r, s=s))) for i in range(2)]
            for t in th: t.start()
            for t in th: t.join()
            assert got[0] == want and got[1] == want, cfg
            for _ in range(12):                                   # (a 2^18 build takes tens of ms: some of these straddle its end)
                assert bn.groth16GenProof(wit, key, r=r, s=s) == want, cfg
            key.wait_tables()
            assert key.load_ms["table_build"] > 0
            assert bn.groth16GenProof(wit, key, r=r, s=s) == want, cfg
            key.free()
            doomed = bn.load_key(sections=sec, wait_tables=False)
            doomed.free()                                         # the destructor waits for the build queue
    finally:
        bn.lib.tune("TABLE_SLAB_LANES", None)


def test_loads_frees_and_proofs_overlap(bn):
    """What the Node suite does, from Python threads (round 4: the background table builds of keys loaded back to back are still
    queued while later keys load and proofs run; an intermediate version whose build scratch came from the stream-ordered
    allocator returned wrong proofs in this pattern while every test that does one thing at a time passed).  One thread keeps
    loading and freeing small reference keys (builds pile up on the context's build queue), two others prove -- on keys whose
    tables are long ready, on keys loaded a moment ago, on a 2^12 circuit -- and every proof must be the golden / closed-form one."""
    import threading
    from wasmsnark_amd import synth
    gold = load_golden("proofs.json")
    names = [n for n in gold if os.path.exists(os.path.join(GOLDEN, "keys", n + ".pkey.bin"))]
    files = {n: (open(os.path.join(GOLDEN, "keys", n + ".pkey.bin"), "rb").read(), open(os.path.join(GOLDEN, "keys", n + ".witness.bin"), "rb").read()) for n in names}
    circ = synth.NativeCircuit(bn.lib, 12, n_public=5, seed=46)
    sec, _ = circ.build_sections()
    cwit = circ.witness_bin()
    r, s = os.urandom(32), os.urandom(32)
    cwant = circ.expected_proof(r, s)
    old = {n: bn.load_key(files[n][0]) for n in names}                  # tables ready
    stop = threading.Event()
    errors = []

    def loader():
        try:
            i = 0
            while not stop.is_set():
                n = names[i % len(names)]
                k = bn.load_key(files[n][0], wait_tables=False)
                if i % 3 == 0:
                    k2 = bn.load_key(sections=sec, wait_tables=False)
                    if bn.groth16GenProof(cwit, k2, r=r, s=s) != cwant: errors.append("fresh 2^12 key")
                    k2.free()
                c = gold[n][i % len(gold[n])]
                if bn.groth16GenProof(files[n][1], k, r=H(c["r"]), s=H(c["s"])) != c["proof"]: errors.append("fresh key " + n)
                k.free()
                i += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def prover(seed):
        try:
            rnd = random.Random(seed)
            for _ in range(600):
                n = rnd.choice(names)
                c = rnd.choice(gold[n])
                if bn.groth16GenProof(files[n][1], old[n], r=H(c["r"]), s=H(c["s"])) != c["proof"]: errors.append("resident key " + n)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=loader)] + [threading.Thread(target=prover, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th[1:]: t.join()
    stop.set()
    th[0].join()
    for k in old.values(): k.free()
    assert not errors, errors[:5]


def test_sections_loader_2p18_equals_file_loader(bn):
    """wsnark_pkey_load_sections (the container for keys beyond proving_key.bin's 4 GiB of u32 offsets: BASELINE
    config 5) against wsnark_pkey_load on the same 2^18 key: same proofs, equal to the closed form; short
    sections are refused."""
    from wasmsnark_amd import WsnarkError, synth
    circ = synth.make_circuit(18, n_public=5, seed=18)
    S = synth.setup(circ, seed=4)
    sec, _ = synth.build_sections(circ, S, bn.mul_base)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    k_file, k_sec = bn.load_key(pkey), bn.load_key(sections=sec)
    assert (k_sec.n_vars, k_sec.n_public, k_sec.domain) == (k_file.n_vars, k_file.n_public, k_file.domain)
    wit = synth.witness_bin(circ)
    r, s = os.urandom(32), os.urandom(32)
    want = synth.expected_proof(circ, S, r, s, bn.mul_base)
    assert bn.groth16GenProof(wit, k_sec, r=r, s=s) == want
    assert bn.groth16GenProof(wit, k_file, r=r, s=s) == want
    for name in ("pointsB2", "pointsH"):
        short = dict(sec)
        short[name] = sec[name][:-64]
        with pytest.raises(WsnarkError):
            bn.load_key(sections=short)
    k_file.free(); k_sec.free()


def test_two_proofs_in_flight_one_key_on_gpu(bn):
    """Lanes: two host threads prove with ONE key handle at the same time (each call holds a lane: its own queues,
    MSM plans, scratch, witness / h buffers), a third waits for a lane; every proof equals the closed form."""
    import threading
    from wasmsnark_amd import synth
    circ = synth.make_circuit(16, n_public=5, seed=61)
    S = synth.setup(circ, seed=6)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    key = bn.load_key(pkey)
    wit = synth.witness_bin(circ)
    jobs = []
    for i in range(3):
        r, s = bytes([i + 1]) * 32, bytes([i + 101]) * 32
        jobs.append((r, s, synth.expected_proof(circ, S, r, s, bn.mul_base)))
    errors = []

    def worker(r, s, want):
        try:
            for _ in range(8):
                if bn.groth16GenProof(wit, key, r=r, s=s) != want:
                    errors.append("proof mismatch")
                if bn.last_blinding() != (r, s):
                    errors.append("last_blinding is not per thread")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


def test_concurrent_callers(bn, orc):
    """Four host threads at once on one context: proofs with two different resident keys, host-pointer MSMs,
    and NTT / CALC_H calls.  The library serialises what shares scratch; every result must still be exact."""
    import threading
    from wasmsnark_amd import synth
    jobs = []
    for logd, seed in ((12, 21), (13, 22)):
        circ = synth.make_circuit(logd, n_public=3, seed=seed)
        S = synth.setup(circ, seed=seed + 100)
        pkey, _ = synth.build_key(circ, S, bn.mul_base)
        r32, s32 = bytes([seed]) * 32, bytes([seed + 1]) * 32
        jobs.append((bn.load_key(pkey), synth.witness_bin(circ), r32, s32, synth.expected_proof(circ, S, r32, s32, bn.mul_base)))
    n = 5000
    rnd = random.Random(5)
    ks = [rnd.randrange(1, orc.R) for _ in range(n)]
    pts = bn.mul_base(1, b"".join(k.to_bytes(32, "little") for k in ks))
    ss = [rnd.randrange(1 << 256) for _ in range(n)]
    sc = b"".join(v.to_bytes(32, "little") for v in ss)
    want_msm = bn.mul_base(1, (sum(a * b for a, b in zip(ss, ks)) % orc.R).to_bytes(32, "little"))
    errors = []

    def prover(key, wit, r32, s32, want):
        try:
            for _ in range(6):
                if bn.groth16GenProof(wit, key, r=r32, s=s32) != want:
                    errors.append("proof mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def msms():
        try:
            for _ in range(12):
                if bn.g1_multiexp(sc, pts)[:64] != want_msm:
                    errors.append("msm mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def transforms():
        try:
            data = rand_fr(random.Random(9), 1 << 12)
            mont = bn.toMontgomeryN(data)
            for _ in range(10):
                if bn.ifft(bn.fft(mont)) != mont:
                    errors.append("ntt round trip mismatch")
            for c in load_golden("calch.json"):
                h = bn.calcH(B64(c["signals"]), B64(c["polsA"]), B64(c["polsB"]), c["nSignals"], c["domain"])
                if h != B64(c["h"]):
                    errors.append("calcH mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = ([threading.Thread(target=prover, args=j) for j in jobs] + [threading.Thread(target=msms)]
          + [threading.Thread(target=transforms)])
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


def test_prove_small_vs_oracle(bn, orc):
    from wasmsnark_amd import synth
    circ = synth.make_circuit(8, n_public=2, seed=88)
    S = synth.setup(circ, seed=5)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    wit = synth.witness_bin(circ)
    r, s = os.urandom(32), os.urandom(32)
    assert bn.groth16GenProof(wit, pkey, r=r, s=s) == orc.groth16_prove(wit, pkey, r, s, workers=8)


def test_c1_example_witness_vs_reference_proof(bn, orc):
    """BASELINE config 1: the reference's example witness (66 232 signals, real circuit value
    distribution: zeros, ones, small values) on a seeded pseudo-key; expected proof = what the
    REFERENCE prover (Node + WASM) produced for the same bytes (tests/gen_golden_c1.py)."""
    import hashlib
    from wasmsnark_amd import synth
    g = load_golden("c1_example.json")
    wit = open(os.path.join(GOLDEN, "example_witness.bin"), "rb").read()
    pkey = synth.pseudo_key(g["n_vars"], g["n_public"], g["domain"], g["key_seed"], bn.mul_base)
    assert hashlib.sha256(pkey).hexdigest() == g["pkey_sha256"]          # same key bytes as the reference saw
    r, s = H(g["r"]), H(g["s"])
    key = bn.load_key(pkey)
    assert bn.groth16GenProof(wit, key, r=r, s=s) == g["proof"]
    assert orc.groth16_prove(wit, pkey, r, s, workers=32) == g["proof"]  # and the oracle agrees at this size too


@pytest.mark.parametrize("logd", [13, 16, 20])
def test_native_dist_prover_world_of_one_on_gpu(bn, logd):
    """wsnark_groth16_prove_dist with a world of one: the same kernels, queues and exchange-buffer layouts as N ranks (the
    exchange itself is the identity), odd and even log2(domain), against the closed form and the one-call prover."""
    import torch
    from wasmsnark_amd import dist as wd, synth
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=40 + logd)
    sec, _ = circ.build_sections()
    wit = circ.witness_bin()
    dev = torch.device("cuda", 0)
    d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    npv = wd.NativeDistProver(bn, sec, device=dev)
    assert npv.key.shard["h_interleave_log"] == logd // 2 and npv.key.table["rows_w"] > 1
    whole = bn.load_key(sections=sec)
    for _ in range(2):
        r, s = os.urandom(32), os.urandom(32)
        got = npv.prove(d_w.data_ptr(), len(wit), r=r, s=s)
        assert got == circ.expected_proof(r, s) == bn.groth16GenProof(wit, whole, r=r, s=s)
    whole.free()
    npv.key.free()


def test_window_shards_sum_to_full_msm_on_gpu(bn, orc):
    # wsnark_g1_msm_windows: the partial sums over the window shards of any world size add up
    rnd = random.Random(321)
    n = 5000
    pts = bn.mul_base(1, b"".join(rnd.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n)))
    sc = _skewed_scalars(rnd, n, orc.R)
    want = orc.g_affine(1, orc.multiexp(1, "workers8", sc, pts, n))
    for world in (2, 8):
        parts = b"".join(bn.g1_multiexp(sc, pts, shard=(rank, world)) for rank in range(world))
        assert bn.g1_sum(parts) == want, world
    assert bn.g1_multiexp(sc, pts) == want


def test_sharded_prove_records_on_gpu(bn):
    """wsnark_groth16_prove_partial / _finish: the 576-byte records of a simulated world of 4 window
    shards combine to the single-GPU proof (and to the toxic-waste closed form)."""
    from wasmsnark_amd import synth
    circ = synth.make_circuit(12, n_public=3, seed=77)
    S = synth.setup(circ, seed=9)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    key = bn.load_key(pkey)
    wit = synth.witness_bin(circ)
    r, s = os.urandom(32), os.urandom(32)
    want = synth.expected_proof(circ, S, r, s, bn.mul_base)
    assert bn.groth16GenProof(wit, key, r=r, s=s) == want
    parts = b"".join(bn.groth16_prove_partial(wit, key, shard=(rank, 4)) for rank in range(4))
    assert len(parts) == 4 * 576
    assert bn.groth16_prove_finish(key, parts, r=r, s=s) == want


@pytest.mark.parametrize("logd", [12, 16])
def test_key_tables_and_plain_sections_give_the_same_proofs_on_gpu(bn, tune, logd):
    """Resident keys are fixed-base window tables by default (wsnark_pkey_table_info: rows x n points per section, one
    bucket set per sum); KEY_TABLE=0 keeps the plain sections and the per-window plans (2 / 3: tables for the hExps / for the
    witness sections alone).  Same proofs from all,
    equal to the closed form; also with the table's bucket set cut into several reduction pieces, with the masked plan
    variants, and through the window shards of a world of 3."""
    from wasmsnark_amd import synth
    circ = synth.make_circuit(logd, n_public=4, seed=100 + logd, style="rows")
    S = synth.setup(circ, seed=5)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    wit = synth.witness_bin(circ)
    r, s = os.urandom(32), os.urandom(32)
    want = synth.expected_proof(circ, S, r, s, bn.mul_base)
    key = bn.load_key(pkey)
    assert key.table["rows_w"] == -(-255 // key.table["c_w"]) > 1 and key.table["c_w"] == min(20, circ.n_vars.bit_length() - 1)
    assert key.table["bytes"] == circ.n_vars * 320 * key.table["rows_w"] + circ.domain * 64 * key.table["rows_h"]
    assert bn.groth16GenProof(wit, key, r=r, s=s) == want
    parts = b"".join(bn.groth16_prove_partial(wit, key, shard=(rank, 3)) for rank in range(3))
    assert bn.groth16_prove_finish(key, parts, r=r, s=s) == want
    key.free()
    for cfg in ({"KEY_TABLE": 0}, {"KEY_TABLE": 2}, {"KEY_TABLE": 3}, {"TABLE_C": 10, "TAIL_BITS": 6, "PROVE_SPARSE": 2}, {"TABLE_C": 13, "MSM_ENTRY64": 1}):
        for k, v in cfg.items():
            tune(bn.lib, k, v)
        key = bn.load_key(pkey)
        assert (key.table["rows_w"] == 1) == (cfg.get("KEY_TABLE") in (0, 2)) and (key.table["rows_h"] == 1) == (cfg.get("KEY_TABLE") in (0, 3))
        assert bn.groth16GenProof(wit, key, r=r, s=s) == want, cfg
        key.free()
        for k in cfg:
            bn.lib.tune(k, None)


@pytest.mark.parametrize("mode", ["table", "plain"])
def test_degenerate_key_points_against_the_oracle_prover_on_gpu(bn, orc, tune, mode):
    """Equal points with equal scalars (doubling inside a bucket), P / -P neighbours, points at infinity in the A, B1, B2
    and hExps sections of the t6 key: the GPU prover against the oracle's restatement of the reference prover, on the
    fixed-base table key and on plain sections (tests/primitives_common.py: degenerate_key_and_witness)."""
    from primitives_common import degenerate_key_and_witness
    if mode == "plain":
        tune(bn.lib, "KEY_TABLE", 0)
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", "t6" + ext), "rb").read()
    key, w = degenerate_key_and_witness(orc, rd(".pkey.bin"), rd(".witness.bin"))
    k = bn.load_key(key)
    assert (k.table["rows_w"] > 1) == (mode == "table")
    for r, s in ((bytes(32), bytes(32)), (bytes(range(1, 33)), bytes(range(101, 133)))):
        assert bn.groth16GenProof(w, k, r=r, s=s) == orc.groth16_prove(w, key, r, s, workers=8)
    k.free()


def test_ntt_2p25_four_pass_roundtrip_and_linearity(bn):
    """2^25 needs four digit passes (middle-digit reversal): round trip, and F(x + y) == F(x) + F(y)
    checked through the evaluation at one point: sum_k F(x)[k] == n * x[0] (DFT of the constant-one vector)."""
    import torch
    sync = torch.cuda.synchronize
    n = 1 << 25
    g = torch.Generator(device="cpu").manual_seed(25)
    x = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, generator=g)
    x[31::32] &= 0x0F
    d = x.cuda()
    y = d.clone()
    sync()
    bn.fft_dev(y.data_ptr(), n, 1)
    bn.fft_dev(y.data_ptr(), n, 1, inverse=False)   # two forward coset transforms ...
    sync()
    z = d.clone()
    sync()
    bn.fft_dev(z.data_ptr(), n, 0)
    bn.fft_dev(z.data_ptr(), n, 0, inverse=True)
    sync()
    assert torch.equal(z, d)                        # ifft(fft(x)) == x
    assert not torch.equal(y, d)


@pytest.mark.parametrize("bits", [6, 11, 16, 20, 22])
def test_four_step_ntt_building_blocks_on_gpu(bn, orc, bits):
    """wsnark_fr_ntt_batch_dev + wsnark_fr_dist_scale_dev through dist_ntt with a world of one (column step, twiddle,
    transpose, row step on ONE GPU): bit-identical to the single-kernel-chain transform wsnark_fr_ntt_dev for odd 0/1,
    forward and inverse (and to the oracle where it is fast).  The N > 1 exchange itself is in tests/test_gpu_multi.py
    (nccl) and tests/test_dist_ntt_gloo.py (CPU)."""
    import torch
    from wasmsnark_amd import dist as wd
    n = 1 << bits
    g = torch.Generator(device="cpu").manual_seed(bits)
    x = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, generator=g)
    x[31::32] &= 0x1F
    d = x.cuda()
    l1, l2 = wd.ntt_layout_split(bits, 1)
    for odd in (0, 1):
        for inverse in (False, True):
            ref = d.clone()
            torch.cuda.synchronize()
            bn.fft_dev(ref.data_ptr(), n, odd, inverse=inverse)
            torch.cuda.synchronize()
            loc = wd.to_interleaved(d, l1, 0, 1).clone()
            y = wd.from_interleaved(wd.dist_ntt(bn, loc, bits, odd=odd, inverse=inverse), l2)
            assert torch.equal(y, ref), (bits, odd, inverse)
            if bits <= 16:
                assert ref.cpu().numpy().tobytes() == orc.fft(x.numpy().tobytes(), n, odd, inverse=inverse)


def test_native_verifier_accepts_gpu_proofs(bn):
    """wsnark_groth16_verify (host pairing) on proofs the GPU just produced -- with library-drawn blinding, so that nothing
    but the verifier vouches for them -- and rejects them for a wrong public input."""
    from wasmsnark_amd import synth
    circ = synth.make_circuit(14, n_public=4, seed=141)
    S = synth.setup(circ, seed=14)
    pkey, vk = synth.build_key(circ, S, bn.mul_base)
    key = bn.load_key(pkey)
    wit = synth.witness_bin(circ)
    pub = synth.public_signals(circ)
    for _ in range(2):
        proof = bn.groth16GenProof(wit, key)
        assert bn.groth16Verify(vk, pub, proof) is True
        assert bn.groth16Verify(vk, [str(int(pub[0]) + 1)] + pub[1:], proof) is False
    key.free()


@pytest.mark.parametrize("logd", [13, 16])
def test_dist_prover_world_of_one_on_gpu(bn, logd):
    """wasmsnark_amd.dist.DistProver with a world of one: CALC_H through the four-step building blocks (eval_ab, fr_mul,
    batched transforms, dist_scale, dist_combine: odd and even log2(domain)), the H sum from the caller's h and H points,
    the other four sums with WSNARK_PARTIAL_SKIP_H -- equal to the closed form and to the one-call prover."""
    import struct
    import torch
    from wasmsnark_amd import dist as wd, synth
    circ = synth.make_circuit(logd, n_public=5, seed=200 + logd)
    S = synth.setup(circ, seed=20)
    pkey, _ = synth.build_key(circ, S, bn.mul_base)
    key = bn.load_key(pkey)
    wit = synth.witness_bin(circ)
    d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    dp = wd.DistProver(bn, key, pkey[struct.unpack_from("<I", pkey, 36)[0]:], device=torch.device("cuda", 0))
    r, s = os.urandom(32), os.urandom(32)
    want = synth.expected_proof(circ, S, r, s, bn.mul_base)
    assert dp.prove(d_w.data_ptr(), len(wit), r=r, s=s) == want
    assert bn.groth16GenProof(wit, key, r=r, s=s) == want
    key.free()


@pytest.mark.gpu
@pytest.mark.parametrize("g", [1, 2])
def test_resident_bases_give_the_per_call_sums(bn, g):
    """wsnark_points_load: the bases resident as fixed-base tables; sums from the handle == sums from the bytes (reference
    src/bn128.js:353-415 semantics), for uniform, skewed and all-zero scalars, from host and from device scalars."""
    import numpy as np
    import torch
    n = 1 << (18 if g == 1 else 16)
    rng = np.random.default_rng(77 + g)
    ks = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); ks[:, 31] &= 0x1F
    pts = bn.mul_base(g, ks.tobytes())
    rp = bn.load_points(g, pts)
    assert rp.n == n and rp.table["rows"] * rp.table["c"] >= 254
    per_call = bn.g1_multiexp if g == 1 else bn.g2_multiexp
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x1F
    skew = sc.copy(); u = rng.random(n); skew[u < 0.07] = 0; skew[(u >= 0.07) & (u < 0.10)] = 0; skew[(u >= 0.07) & (u < 0.10), 0] = 1
    skew[u > 0.9, 4:] = 0
    for s_ in (sc, skew, np.zeros_like(sc)):
        want = per_call(s_.tobytes(), pts)
        assert rp.multiexp(s_.tobytes()) == want
        d = torch.from_numpy(s_.reshape(-1)).cuda()
        assert rp.multiexp_dev(d.data_ptr(), n) == want
    with pytest.raises(Exception):
        rp.multiexp(sc[: n // 2].tobytes())          # a handle sums over the whole set
    rp.free()
