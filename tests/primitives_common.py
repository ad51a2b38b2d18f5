"""The reference's own primitive vectors (tests/golden/fields.json, groups.json -- harvested from
/root/reference test/f1.js:296-400 and test/bn128.js:84-185 by oracle/ref_harness/gen_golden.js) run through
the product's field and curve arithmetic via wsnark_selftest_field / wsnark_selftest_curve.

Shared by tests/test_gpu_primitives.py (-m gpu: the real device code on an MI355X) and
tests/test_emul_primitives.py (CPU: the same sources under the thread emulator).  Everything is bit-exact."""
import ctypes as C
import random

import pytest

from conftest import load_golden

H = bytes.fromhex

MUL, SQR, ADD, SUB, NEG, TOMONT, FROMMONT, SUB_WEAK, ADD_LAZY_MUL, NEG_WEAK_MUL, MUL2ADD, MULSUB2, EQ, EQ_WEAK, INVERSE, SQR_WEAK, MUL_WEAK_A, MULSUB2_WEAK_B = range(18)
FIELD_IMPLS = {0: "radix-2^29 device field", 1: "saturated device field", 2: "host field"}
CURVE_IMPLS = {1: (0, 1, 2, 3), 2: (0, 1, 2, 4)}    # 3: G1 tail variant (inlined products); 4: G2 on lane pairs (reduction-tail kernels)


def st_field(bn, which, impl, op, a_list, b_list):
    sz = 64 if which == 2 else 32
    n = len(a_list)
    a, b = b"".join(a_list), b"".join(b_list)
    assert len(a) == n * sz and len(b) == n * sz
    out = (C.c_uint8 * (n * sz))()
    bn.lib.check(bn.lib.c.wsnark_selftest_field(which, impl, op, a, b, out, n))
    o = bytes(out)
    return [o[i * sz:(i + 1) * sz] for i in range(n)]


def st_curve(bn, g, impl, op, p_list, q_list):
    sz = 96 if g == 1 else 192
    n = len(p_list)
    p, q = b"".join(p_list), b"".join(q_list)
    assert len(p) == n * sz and len(q) == n * sz
    out = (C.c_uint8 * (n * sz))()
    bn.lib.check(bn.lib.c.wsnark_selftest_curve(g, impl, op, p, q, out, n))
    o = bytes(out)
    return [o[i * sz:(i + 1) * sz] for i in range(n)]


def _le(v):
    return int(v).to_bytes(32, "little")


def _int(b):
    return int.from_bytes(b, "little")


def check_base_field(bn, fname, which, impl):
    """f1m_* / frm_* vectors of the reference edge grid, plus the lazy / fused forms the kernels really use."""
    G = load_golden("fields.json")
    p = int(G["q" if fname == "fq" else "r"])
    rinv = pow(1 << 256, p - 2, p)
    F = G["fields"][fname]
    un, bi = F["unary"], F["binary"]
    ua = [H(c["a"]) for c in un]
    assert st_field(bn, which, impl, SQR, ua, ua) == [H(c["square"]) for c in un]
    assert st_field(bn, which, impl, NEG, ua, ua) == [H(c["neg"]) for c in un]
    assert st_field(bn, which, impl, TOMONT, ua, ua) == [H(c["toMontgomery"]) for c in un]
    assert st_field(bn, which, impl, FROMMONT, ua, ua) == [H(c["fromMontgomery"]) for c in un]
    # f1m_inverse vectors: the host's inversion (impl 2) and the DEVICE's Fermat inversion on both device fields (impl 0: the one the
    # table build's normalisation runs, msm_table_norm_kernel; src/build_f1m.js:772-782 gives the same unique value)
    inv = [c for c in un if "inverse" in c]
    assert inv
    ia = [H(c["a"]) for c in inv]
    assert st_field(bn, which, impl, INVERSE, ia, ia) == [H(c["inverse"]) for c in inv]
    rnd_inv = random.Random(900 + which * 10 + impl)
    xs = [rnd_inv.randrange(1, p) for _ in range(64)]
    assert st_field(bn, which, impl, INVERSE, [_le(x) for x in xs], [_le(x) for x in xs]) == [_le(pow(x, p - 2, p) * pow(1 << 256, 2, p) % p) for x in xs]
    ba, bb = [H(c["a"]) for c in bi], [H(c["b"]) for c in bi]
    assert st_field(bn, which, impl, MUL, ba, bb) == [H(c["mul"]) for c in bi]
    assert st_field(bn, which, impl, ADD, ba, bb) == [H(c["add"]) for c in bi]
    assert st_field(bn, which, impl, SUB, ba, bb) == [H(c["sub"]) for c in bi]
    # the forms the kernels use around their products: same values as the strict ones
    assert st_field(bn, which, impl, SUB_WEAK, ba, bb) == [H(c["sub"]) for c in bi]
    # edge grid x edge grid (the reference's set: 0, 1, 2, p-1, p-2, (p-1)/2 +- k, ...) plus seeded randoms,
    # expected values from plain integer arithmetic on the Montgomery representatives
    rnd = random.Random(which * 10 + impl)
    edge = sorted({_int(x) for x in ua} | {0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2})
    pairs = [(x, y) for x in edge for y in edge] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(256)]
    pa, pb = [_le(x) for x, _ in pairs], [_le(y) for _, y in pairs]
    assert st_field(bn, which, impl, MUL, pa, pb) == [_le(x * y * rinv % p) for x, y in pairs]
    assert st_field(bn, which, impl, ADD, pa, pb) == [_le((x + y) % p) for x, y in pairs]
    assert st_field(bn, which, impl, SUB, pa, pb) == [_le((x - y) % p) for x, y in pairs]
    assert st_field(bn, which, impl, SUB_WEAK, pa, pb) == [_le((x - y) % p) for x, y in pairs]
    assert st_field(bn, which, impl, ADD_LAZY_MUL, pa, pb) == [_le((x + y) * y * rinv % p) for x, y in pairs]
    assert st_field(bn, which, impl, NEG_WEAK_MUL, pa, pb) == [_le((p - x) * y * rinv % p) for x, y in pairs]
    assert st_field(bn, which, impl, MUL2ADD, pa, pb) == [_le((x * x + y * y) * rinv % p) for x, y in pairs]
    assert st_field(bn, which, impl, MULSUB2, pa, pb) == [_le(((x - y) * x - y * x) * rinv % p) for x, y in pairs]
    assert st_field(bn, which, impl, SQR_WEAK, pa, pb) == [_le((x - y) * (x - y) * rinv % p) for x, y in pairs]
    assert st_field(bn, which, impl, MUL_WEAK_A, pa, pb) == [_le((x - y) * y * rinv % p) for x, y in pairs]
    assert st_field(bn, which, impl, MULSUB2_WEAK_B, pa, pb) == [_le((x * (x - y) - y * x) * rinv % p) for x, y in pairs]
    flags = [_le(1 if x == y else 0) for x, y in pairs]
    assert st_field(bn, which, impl, EQ, pa, pb) == flags
    assert st_field(bn, which, impl, EQ_WEAK, pa, pb) == flags


def check_fq2(bn, impl):
    """f2m_mul / square / inverse vectors of the reference, plus add / sub / neg / fused a*b - c*d against integers."""
    G = load_golden("fields.json")
    q = int(G["q"])
    rinv = pow(1 << 256, q - 2, q)
    cs = G["fq2"]
    a, b = [H(c["a"]) for c in cs], [H(c["b"]) for c in cs]
    assert st_field(bn, 2, impl, MUL, a, b) == [H(c["mul"]) for c in cs]
    assert st_field(bn, 2, impl, SQR, a, a) == [H(c["square"]) for c in cs]
    if impl == 2:
        inv = [c for c in cs if "inverse" in c]
        assert inv
        ia = [H(c["a"]) for c in inv]
        assert st_field(bn, 2, impl, INVERSE, ia, ia) == [H(c["inverse"]) for c in inv]
    rnd = random.Random(200 + impl)
    edge = [0, 1, 2, q - 1, q - 2, (q - 1) // 2, (q + 1) // 2]
    vals = [(x, y) for x in edge for y in edge] + [(rnd.randrange(q), rnd.randrange(q)) for _ in range(64)]
    pairs = [(rnd.choice(vals), rnd.choice(vals)) for _ in range(400)] + [(v, v) for v in vals[:60]]
    enc = lambda v: _le(v[0]) + _le(v[1])
    pa, pb = [enc(x) for x, _ in pairs], [enc(y) for _, y in pairs]
    mul = lambda x, y: ((x[0] * y[0] - x[1] * y[1]) * rinv % q, (x[0] * y[1] + x[1] * y[0]) * rinv % q)   # u^2 = -1
    sub = lambda x, y: ((x[0] - y[0]) % q, (x[1] - y[1]) % q)
    assert st_field(bn, 2, impl, MUL, pa, pb) == [enc(mul(x, y)) for x, y in pairs]
    assert st_field(bn, 2, impl, SQR, pa, pa) == [enc(mul(x, x)) for x, _ in pairs]
    assert st_field(bn, 2, impl, ADD, pa, pb) == [enc(((x[0] + y[0]) % q, (x[1] + y[1]) % q)) for x, y in pairs]
    assert st_field(bn, 2, impl, SUB, pa, pb) == [enc(sub(x, y)) for x, y in pairs]
    assert st_field(bn, 2, impl, NEG, pa, pa) == [enc(((-x[0]) % q, (-x[1]) % q)) for x, _ in pairs]
    want = [enc(sub(mul(sub(x, y), x), mul(y, x))) for x, y in pairs]
    assert st_field(bn, 2, impl, MULSUB2, pa, pb) == want
    # uncorrected differences where the G2 formulas put them: operand of the squaring, first operand of the product,
    # second operand of the fused a*b - c*d
    assert st_field(bn, 2, impl, SUB_WEAK, pa, pb) == [enc(sub(x, y)) for x, y in pairs]
    assert st_field(bn, 2, impl, SQR_WEAK, pa, pb) == [enc(mul(sub(x, y), sub(x, y))) for x, y in pairs]
    assert st_field(bn, 2, impl, MUL_WEAK_A, pa, pb) == [enc(mul(sub(x, y), y)) for x, y in pairs]
    assert st_field(bn, 2, impl, MULSUB2_WEAK_B, pa, pb) == [enc(sub(mul(x, sub(x, y)), mul(y, x))) for x, y in pairs]
    flag = lambda v: _le(v) + bytes(32)
    assert st_field(bn, 2, impl, EQ, pa, pb) == [flag(1 if x == y else 0) for x, y in pairs]
    assert st_field(bn, 2, impl, EQ_WEAK, pa, pb) == [flag(1 if x == y else 0) for x, y in pairs]


def check_group(bn, orc, g, impl):
    """g1m_* / g2m_* vectors: generic, P+P, same point with different z, P-P, infinity operands (reference branches
    src/build_curve_jacobian_a0.js:322-356), through the full addition, the doubling and the MIXED addition."""
    G = load_golden("groups.json")["g%d" % g]
    cs = G["cases"]
    p, q = [H(c["p"]) for c in cs], [H(c["q"]) for c in cs]
    labels = [c["label"] for c in cs]
    got = st_curve(bn, g, impl, 0, p, q)
    assert got == [H(c["add_affine"]) for c in cs], [l for l, x, c in zip(labels, got, cs) if x != H(c["add_affine"])]
    assert st_curve(bn, g, impl, 0, q, p) == [H(c["add_affine"]) for c in cs]          # commutes
    assert st_curve(bn, g, impl, 1, p, p) == [H(c["double_affine"]) for c in cs]
    assert st_curve(bn, g, impl, 2, p, p) == [orc.g_affine(g, H(c["neg"])) for c in cs]
    assert st_curve(bn, g, impl, 3, p, p) == [H(c["p_affine"]) for c in cs]
    qa = [orc.g_affine(g, x) for x in q]                                              # the key's points are affine
    got = st_curve(bn, g, impl, 4, p, qa)
    assert got == [H(c["add_affine"]) for c in cs], [l for l, x, c in zip(labels, got, cs) if x != H(c["add_affine"])]
    assert st_curve(bn, g, impl, 5, p, qa) == [orc.g_affine(g, orc.g_add(g, x, orc.g_neg(g, y))) for x, y in zip(p, q)]
    # the accumulation loop's lazy form: two mixed additions with x kept wide in between, then narrowed
    assert st_curve(bn, g, impl, 6, p, qa) == [orc.g_affine(g, orc.g_add(g, orc.g_add(g, x, y), y)) for x, y in zip(p, q)]
    assert st_curve(bn, g, impl, 7, p, qa) == [orc.g_affine(g, x) for x in p]
    # infinity is always written as (0, 1, 0) and P + (-P) lands there
    zero = orc.g_affine(g, orc.g_zero(g))
    assert st_curve(bn, g, impl, 0, p, [orc.g_neg(g, x) for x in p]) == [zero] * len(p)
    assert st_curve(bn, g, impl, 5, p, [orc.g_affine(g, x) for x in p]) == [zero] * len(p)
    # group law of test/bn128.js:84-134: 4G by additions == by doublings; 10G == timesScalar(G, 10)
    gen = H(G["gen"])
    two = st_curve(bn, g, impl, 1, [gen], [gen])[0]
    four_d = st_curve(bn, g, impl, 1, [two], [two])[0]
    acc = gen
    multiples = [gen]
    for _ in range(9):
        acc = st_curve(bn, g, impl, 0, [acc], [gen])[0]
        multiples.append(acc)
    assert multiples[3] == four_d
    ten = [c for c in G["times_scalar"] if _int(H(c["scalar"])) == 10 and c["bytes"] == 32]
    assert ten and multiples[9] == H(ten[0]["affine"])
    # g{1,2}m_timesScalar itself (src/build_timesscalar.js:20-80; the reference calls it with 32- AND 64-byte scalars,
    # src/bn128.js:672-702): every reference vector, incl. the 64-byte one, through this implementation's double-and-add
    if impl != 4:
        sz = 96 if g == 1 else 192
        ts = G["times_scalar"]
        assert any(c["bytes"] == 64 for c in ts)
        ops = [H(c["scalar"]).ljust(64, b"\0") + bytes([c["bytes"]]) + bytes(sz - 65) for c in ts]
        assert st_curve(bn, g, impl, 8, [gen] * len(ts), ops) == [H(c["affine"]) for c in ts]
    mixed = gen
    for _ in range(9):
        mixed = st_curve(bn, g, impl, 4, [mixed], [gen])[0]
    assert mixed == multiples[9]
    # seeded random Jacobian points with non-unit z (oracle timesScalar outputs) against the oracle's own add / double
    rnd = random.Random(40 + g)
    n = 24 if g == 1 else 10
    P = [orc.g_times_scalar(g, gen, _le(rnd.randrange(1, orc.R))) for _ in range(n)]
    Q = [orc.g_times_scalar(g, gen, _le(rnd.randrange(1, orc.R))) for _ in range(n)]
    assert st_curve(bn, g, impl, 0, P, Q) == [orc.g_affine(g, orc.g_add(g, x, y)) for x, y in zip(P, Q)]
    assert st_curve(bn, g, impl, 1, P, P) == [orc.g_affine(g, orc.g_double(g, x)) for x in P]
    Qa = [orc.g_affine(g, y) for y in Q]
    assert st_curve(bn, g, impl, 4, P, Qa) == [orc.g_affine(g, orc.g_add(g, x, y)) for x, y in zip(P, Q)]
    assert st_curve(bn, g, impl, 4, P, [orc.g_affine(g, x) for x in P]) == [orc.g_affine(g, orc.g_double(g, x)) for x in P]
    assert st_curve(bn, g, impl, 6, P, Qa) == [orc.g_affine(g, orc.g_add(g, orc.g_add(g, x, y), y)) for x, y in zip(P, Q)]
    assert st_curve(bn, g, impl, 7, P, Qa) == [orc.g_affine(g, x) for x in P]
    # p + q + q with p = -q: the first lazy addition lands on infinity, the second restarts from the affine point
    assert st_curve(bn, g, impl, 6, [orc.g_neg(g, y) for y in Qa], Qa) == Qa


def check_group_pair_g1(bn, orc, g=1, impl=5):
    """The lane-split curves of the reduction tails (csrc/curve_pair.h): g = 1, impl 5 -- one G1 point on two lanes; g = 2, impl 6 -- one
    G2 point on four lanes; seven product steps per addition.  The reference's g{1,2}m_add / g{1,2}m_double vectors -- generic, P + P
    (the doubling branch), the same point with another z, P - P, infinity operands (src/build_curve_jacobian_a0.js:322-356) -- and
    seeded random Jacobian operands."""
    G = load_golden("groups.json")["g%d" % g]
    cs = G["cases"]
    p, q = [H(c["p"]) for c in cs], [H(c["q"]) for c in cs]
    labels = [c["label"] for c in cs]
    got = st_curve(bn, g, impl, 0, p, q)
    assert got == [H(c["add_affine"]) for c in cs], [l for l, x, c in zip(labels, got, cs) if x != H(c["add_affine"])]
    assert st_curve(bn, g, impl, 0, q, p) == [H(c["add_affine"]) for c in cs]
    assert st_curve(bn, g, impl, 1, p, p) == [H(c["double_affine"]) for c in cs]
    assert st_curve(bn, g, impl, 0, p, p) == [H(c["double_affine"]) for c in cs]            # P + P takes the doubling branch
    assert st_curve(bn, g, impl, 3, p, p) == [H(c["p_affine"]) for c in cs]
    zero = orc.g_affine(g, orc.g_zero(g))
    assert st_curve(bn, g, impl, 0, p, [orc.g_neg(g, x) for x in p]) == [zero] * len(p)
    gen = H(G["gen"])
    acc, multiples = gen, [gen]
    for _ in range(9):
        acc = st_curve(bn, g, impl, 0, [acc], [gen])[0]
        multiples.append(acc)
    ten = [c for c in G["times_scalar"] if _int(H(c["scalar"])) == 10 and c["bytes"] == 32]
    assert ten and multiples[9] == H(ten[0]["affine"])
    rnd = random.Random(77)
    P = [orc.g_times_scalar(g, gen, _le(rnd.randrange(1, orc.R))) for _ in range(33 if g == 1 else 9)]      # (an odd count: the last pair of lanes alone in its quad)
    Q = [orc.g_times_scalar(g, gen, _le(rnd.randrange(1, orc.R))) for _ in range(33 if g == 1 else 9)]
    assert st_curve(bn, g, impl, 0, P, Q) == [orc.g_affine(g, orc.g_add(g, x, y)) for x, y in zip(P, Q)]
    assert st_curve(bn, g, impl, 1, P, P) == [orc.g_affine(g, orc.g_double(g, x)) for x in P]
    # mixed in one launch: generic, doubling, inverse and infinity operands in NEIGHBOURING lane pairs (the branches are per pair)
    mixP = [P[0], P[1], P[2], orc.g_zero(g), P[4]]
    mixQ = [Q[0], P[1], orc.g_neg(g, P[2]), Q[3], orc.g_zero(g)]
    assert st_curve(bn, g, impl, 0, mixP, mixQ) == [orc.g_affine(g, orc.g_add(g, x, y)) for x, y in zip(mixP, mixQ)]
    with pytest.raises(Exception):
        st_curve(bn, g, impl, 4, p, q)                                                       # (the tails need no mixed addition)


def degenerate_key_and_witness(orc, pkey, wit):
    """The t6 key with degenerate point sections (A, B1, B2, hExps): eight copies of one point, P / -P neighbours, points
    at infinity (x = 0) -- and a witness with EQUAL scalars on the equal points, so that equal digits of equal points meet
    in one bucket (the accumulation's doubling case, and sums that pass through infinity)."""
    import struct
    key = bytearray(pkey)
    nv, npub, dom, pA_, pB_, pA, pB1, pB2, pC, pH = struct.unpack("<10I", pkey[:40])
    Q = orc.Q

    def neg_y(pt, fsz):               # (x, y) -> (x, -y) in the key's Montgomery encoding; fsz = 32 (G1) or 64 (G2: two components)
        out = bytearray(pt)
        for o in range(fsz, 2 * fsz, 32):
            y = int.from_bytes(pt[o:o + 32], "little")
            out[o:o + 32] = ((Q - y) % Q).to_bytes(32, "little")
        return bytes(out)

    for off, esz, n in ((pA, 64, nv), (pB1, 64, nv), (pB2, 128, nv), (pH, 64, dom)):
        P = bytes(key[off + 5 * esz: off + 6 * esz])
        for i in range(6, 14):                                   # eight copies of point 5
            key[off + i * esz: off + (i + 1) * esz] = P
        for i in range(14, 20, 2):                               # P, -P, P, -P, ...
            key[off + i * esz: off + (i + 1) * esz] = P
            key[off + (i + 1) * esz: off + (i + 2) * esz] = neg_y(P, esz // 2)
        for i in range(20, 24):                                  # infinity: x == 0 whatever y is
            key[off + i * esz: off + i * esz + esz // 2] = bytes(esz // 2)
    w = bytearray(wit)
    for i in range(6, 20):
        w[i * 32:(i + 1) * 32] = w[5 * 32:6 * 32]
    return bytes(key), bytes(w)


def check_msm_corner_case_buckets(bn, orc, g, cases=24, seed=5):
    """Every pair of a sum carries the SAME scalar, so each window's bucket holds all the points, in whatever order the grouping pass
    left them: tasks whose first entry is infinity, whose second entry equals or cancels the first (the affine + affine step of the
    accumulation's fast loop must hand both to the generic loop: doubling, infinity), duplicates and P / -P pairs further on, sums
    that pass through infinity and go on.  Against the oracle's restatement of the reference's multiexp (src/build_multiexp.js:498-744),
    whose additions take the branches of src/build_curve_jacobian_a0.js:322-356."""
    rnd = random.Random(seed + g)
    gen = bytes.fromhex(load_golden("groups.json")["g%d" % g]["gen"])
    sz = 64 if g == 1 else 128
    aff = lambda p: orc.g_affine(g, p)[:sz]
    base = []
    for k in (3, 5, 11):
        p = orc.g_times_scalar(g, gen, k.to_bytes(32, "little"))
        base += [aff(p), aff(orc.g_neg(g, p))]
    inf = bytes(sz)
    # hand-made orders first (the grouping pass keeps short lists in input order often enough), then random multisets
    fixed = [[inf, base[0], base[0]], [base[0], base[0], base[2]], [base[0], base[1], base[2]], [base[0], inf, base[1], inf, base[0]],
             [inf, inf, base[4]], [base[2], base[2], base[2], base[3], base[3]], [base[0], base[1], base[0], base[1]], [inf]]
    for i in range(cases):
        pts = fixed[i] if i < len(fixed) else [rnd.choice(base + [inf]) for _ in range(rnd.randrange(2, 14))]
        n = len(pts)
        s = rnd.choice([1, 2, 0x10001, (1 << 255) + 12345, rnd.randrange(1 << 256)])
        sc = s.to_bytes(32, "little") * n
        p = b"".join(pts)
        got = bn.g1_multiexp(sc, p) if g == 1 else bn.g2_multiexp(sc, p)
        want = orc.g_affine(g, orc.multiexp(g, "multiexp2" if g == 1 else "multiexp", sc, p, n))
        assert got == want, (g, i, n, hex(s))
