"""Pins the CPU oracle (oracle/bn128_oracle.c) against vectors produced by the
reference itself (oracle/ref_harness/gen_golden.js ran the reference's WASM).
SURVEY.md section 8(c): the reference holds no numeric KATs for this path, so
reference-generated outputs are the golden vectors."""
import base64

import pytest

from conftest import load_golden

H = bytes.fromhex
B64 = base64.b64decode


def test_constants(orc):
    # SURVEY.md section 8 constants (computed from build_bn128.js:19-20, build_f1m.js:37-38,255)
    m, R, R2, np64 = orc.constants(0)
    assert m == orc.Q
    assert R == 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d
    assert R2 == 0x06d89f71cab8351f47ab1eff0a417ff6b5e71911d44501fbf32cfc5b538afa89
    assert np64 == 0x87d20782e4866389
    m, R, R2, np64 = orc.constants(1)
    assert m == orc.R
    assert R == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert R2 == 0x0216d0b17f4e44a58c49833d53bb808553fe3ab1e35c59e31bb8e645ae216da7
    assert np64 == 0xc2e1f593efffffff


@pytest.mark.parametrize("fname,which", [("fq", 0), ("fr", 1)])
def test_field_vectors(orc, fname, which):
    g = load_golden("fields.json")["fields"][fname]
    for c in g["unary"]:
        a = H(c["a"])
        assert orc.f_un("square", which, a) == H(c["square"])
        assert orc.f_un("neg", which, a) == H(c["neg"])
        assert orc.f_un("to_mont", which, a) == H(c["toMontgomery"])
        assert orc.f_un("from_mont", which, a) == H(c["fromMontgomery"])
        if "inverse" in c:
            assert orc.f_un("inverse", which, a) == H(c["inverse"])
    for c in g["binary"]:
        a, b = H(c["a"]), H(c["b"])
        assert orc.f_bin("mul", which, a, b) == H(c["mul"])
        assert orc.f_bin("add", which, a, b) == H(c["add"])
        assert orc.f_bin("sub", which, a, b) == H(c["sub"])


def test_to_montgomery_11(orc):
    # reference test/f1.js:355-372: toMontgomery(11) == 11 * 2^256 mod r
    out = orc.f_un("to_mont", 1, (11).to_bytes(32, "little"))
    assert int.from_bytes(out, "little") == (11 << 256) % orc.R


def test_fq2_vectors(orc):
    for c in load_golden("fields.json")["fq2"]:
        a, b = H(c["a"]), H(c["b"])
        assert orc.f2_mul(a, b) == H(c["mul"])
        assert orc.f2_square(a) == H(c["square"])
        if "inverse" in c:
            assert orc.f2_inverse(a) == H(c["inverse"])


@pytest.mark.parametrize("g", [1, 2])
def test_group_vectors(orc, g):
    G = load_golden("groups.json")["g%d" % g]
    for c in G["cases"]:
        p, q = H(c["p"]), H(c["q"])
        # the oracle restates the reference formulas, so even the Jacobian
        # representation must be bit-identical
        assert orc.g_add(g, p, q) == H(c["add"]), c["label"]
        assert orc.g_double(g, p) == H(c["double"]), c["label"]
        assert orc.g_neg(g, p) == H(c["neg"]), c["label"]
        assert orc.g_affine(g, H(c["add"])) == H(c["add_affine"]), c["label"]
        assert orc.g_affine(g, p) == H(c["p_affine"]), c["label"]
        assert orc.g_is_zero(g, H(c["add"])) == (1 if c["add_is_zero"] else 0), c["label"]
        assert orc.g_eq(g, p, q) == c["eq"], c["label"]
    gen = H(G["gen"])
    for c in G["times_scalar"]:
        out = orc.g_times_scalar(g, gen, H(c["scalar"]))
        assert orc.g_affine(g, out) == H(c["affine"])


def test_group_laws(orc):
    # reference test/bn128.js:84-185: 4G by adds == by doubles, P-P=0, 10G, r*G == 0
    for g in (1, 2):
        gen = H(load_golden("groups.json")["g%d" % g]["gen"])
        d = orc.g_double(g, orc.g_double(g, gen))
        a = orc.g_add(g, orc.g_add(g, orc.g_add(g, gen, gen), gen), gen)
        assert orc.g_eq(g, d, a) == 1
        assert orc.g_is_zero(g, orc.g_add(g, gen, orc.g_neg(g, gen))) == 1
        ten = gen
        for _ in range(9):
            ten = orc.g_add(g, ten, gen)
        assert orc.g_eq(g, ten, orc.g_times_scalar(g, gen, (10).to_bytes(32, "little"))) == 1
        assert orc.g_is_zero(g, orc.g_times_scalar(g, gen, orc.R.to_bytes(32, "little"))) == 1


@pytest.mark.parametrize("g", [1, 2])
def test_msm_vectors(orc, g):
    for c in load_golden("msm.json")["g%d" % g]:
        s, p, n = B64(c["scalars"]), B64(c["points"]), c["n"]
        if c["flavour"] == "accumulate_into_3G":
            gen = H(load_golden("groups.json")["g%d" % g]["gen"])
            acc = orc.g_times_scalar(g, gen, (3).to_bytes(32, "little"))
            out = orc.multiexp(g, "multiexp2" if g == 1 else "multiexp", s, p, n, acc=acc)
            assert orc.g_affine(g, out) == H(c["acc_affine"])
            continue
        if g == 1:
            assert orc.g_affine(1, orc.multiexp(1, "multiexp2", s, p, n)) == H(c["multiexp2_affine"]), (n, c["flavour"])
        assert orc.g_affine(g, orc.multiexp(g, "multiexp", s, p, n)) == H(c["multiexp_affine"]), (n, c["flavour"])
        if "host_affine" in c:
            assert orc.g_affine(g, orc.multiexp(g, "workers8", s, p, n)) == H(c["host_affine"]), (n, c["flavour"])
            # reference test/bn128_prover.js:9-49: multiexp == multiexp2
            assert c["host_affine"] == c["multiexp_affine"]


def test_msm_kat_14G(orc):
    # MSM([1,2,3],[G,2G,3G]) = 14G
    gen = H(load_golden("groups.json")["g1"]["gen"])
    pts = b""
    for k in (1, 2, 3):
        pts += orc.g_affine(1, orc.g_times_scalar(1, gen, k.to_bytes(32, "little")))[:64]
    sc = b"".join(k.to_bytes(32, "little") for k in (1, 2, 3))
    out = orc.g_from_mont(1, orc.g_affine(1, orc.multiexp(1, "multiexp2", sc, pts, 3)))
    assert int.from_bytes(out[:32], "little") == 9836339169314901400584090930519505895878753154116006108033708428907043344230
    assert int.from_bytes(out[32:64], "little") == 2085718088180884207082818799076507077917184375787335400014805976331012093279


def test_fft_vectors(orc):
    G = load_golden("fft.json")
    for c in G["cases"]:
        x, n = B64(c["input_mont"]), c["n"]
        assert orc.fft(x, n, 0) == B64(c["fft0"])
        assert orc.fft(x, n, 1) == B64(c["fft1"])
        if c["ifft0"] is None:
            with pytest.raises(ValueError):
                orc.fft(x, n, 0, inverse=True)
        else:
            assert orc.fft(x, n, 0, inverse=True) == B64(c["ifft0"])
            assert orc.fft(x, n, 1, inverse=True) == B64(c["ifft1"])
        if "fft0_plain" in c:
            got = orc.from_mont_n(orc.fft(x, n, 0))
            assert got == H(c["fft0_plain"])
            vals = [int.from_bytes(got[i:i + 32], "little") for i in range(0, 128, 32)]
            # SURVEY.md section 8 KAT: fft([0,1,2,3]) = [6, ..., r-2, ...]
            assert vals[0] == 6 and vals[2] == orc.R - 2
            assert vals[3] == 21888242871839275213430563804664787403465736456640143535824009919738970926185
    for t in G["traps"]:
        assert t["traps"] is True
        with pytest.raises(ValueError):
            orc.fft(b"\0" * (32 * max(t["n"], 1)), t["n"], 0)


def test_fft_properties(orc):
    # reference test/fft.js:16-121: ifft(fft(x)) == x; odd-coset interleave == size-2N fft of zero-padded input
    import random
    rnd = random.Random(5)
    n = 256
    x = b"".join(rnd.randrange(orc.R).to_bytes(32, "little") for _ in range(n))
    xm = orc.to_mont_n(x)
    assert orc.fft(orc.fft(xm, n, 0), n, 0, inverse=True) == xm
    e, o = orc.fft(xm, n, 0), orc.fft(xm, n, 1)
    inter = b"".join(e[i * 32:(i + 1) * 32] + o[i * 32:(i + 1) * 32] for i in range(n))
    assert orc.fft(xm + b"\0" * (32 * n), 2 * n, 0) == inter


def test_calc_h_vectors(orc):
    for c in load_golden("calch.json"):
        h = orc.calc_h(B64(c["signals"]), B64(c["polsA"]), B64(c["polsB"]), c["nSignals"], c["domain"])
        assert h == B64(c["h"]), (c["nSignals"], c["domain"])
