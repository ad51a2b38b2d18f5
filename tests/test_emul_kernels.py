"""CPU-only checks of the HIP kernel sources (run under the thread emulator in tests/emul)
against the pinned oracle and the reference-generated golden vectors.  These do NOT replace the
GPU parity tests (tests/test_gpu_parity.py); they exist so index-math bugs are caught here."""
import base64
import random

import pytest

from conftest import load_golden
from emul_util import emul_bn128

H = bytes.fromhex
B64 = base64.b64decode


@pytest.fixture(scope="module")
def bn():
    return emul_bn128()


def test_to_from_montgomery(bn, orc):
    rnd = random.Random(1)
    x = b"".join(rnd.randrange(orc.R).to_bytes(32, "little") for _ in range(300))
    assert bn.toMontgomeryN(x) == orc.to_mont_n(x)
    assert bn.fromMontgomeryN(x) == orc.from_mont_n(x)


def test_fft_golden(bn):
    for c in load_golden("fft.json")["cases"]:
        x = B64(c["input_mont"])
        if c["n"] == 1:
            assert bn.fft(x, 0) == B64(c["fft0"])
            with pytest.raises(Exception):
                bn.ifft(x, 0)
            continue
        assert bn.fft(x, 0) == B64(c["fft0"]), c["n"]
        assert bn.fft(x, 1) == B64(c["fft1"]), c["n"]
        assert bn.ifft(x, 0) == B64(c["ifft0"]), c["n"]
        assert bn.ifft(x, 1) == B64(c["ifft1"]), c["n"]


@pytest.mark.parametrize("bits", [1, 2, 3, 5, 10, 11, 12, 13])
def test_fft_vs_oracle_multi_pass(bn, orc, bits):
    # bits <= 10: single LDS pass; 11..16: two passes (exercises twiddles + digit reversal)
    n = 1 << bits
    rnd = random.Random(bits)
    x = orc.to_mont_n(b"".join(rnd.randrange(orc.R).to_bytes(32, "little") for _ in range(n)))
    for odd in (0, 1):
        assert bn.fft(x, odd) == orc.fft(x, n, odd), (bits, odd)
        assert bn.ifft(x, odd) == orc.fft(x, n, odd, inverse=True), (bits, odd)


@pytest.mark.parametrize("bits", [4, 9, 10, 12])
def test_fft_extreme_values_keep_the_lazy_bounds(bn, orc, bits):
    """The butterflies keep their sums uncorrected (values in [0, 4p) in LDS, folded back from [0, 16p)): vectors
    made of the largest residues -- r - 1 everywhere, r - 1 / 0 / 1 patterns -- drive every sum to its bound."""
    n = 1 << bits
    top = (orc.R - 1).to_bytes(32, "little")
    pats = [top * n,
            b"".join(top if (i & 1) else (0).to_bytes(32, "little") for i in range(n)),
            b"".join((orc.R - 1 - (i % 3)).to_bytes(32, "little") for i in range(n)),
            b"".join(top if i < n // 2 else (1).to_bytes(32, "little") for i in range(n))]
    for plain in pats:
        for x in (plain, orc.to_mont_n(plain)):      # the same bytes read as Montgomery forms, and their Montgomery images
            for odd in (0, 1):
                assert bn.fft(x, odd) == orc.fft(x, n, odd), (bits, odd)
                assert bn.ifft(x, odd) == orc.fft(x, n, odd, inverse=True), (bits, odd)


def test_fft_three_pass(bn, orc):
    n = 1 << 17
    rnd = random.Random(17)
    x = orc.to_mont_n(b"".join(rnd.randrange(orc.R).to_bytes(32, "little") for _ in range(n)))
    assert bn.fft(x, 1) == orc.fft(x, n, 1)
    assert bn.ifft(x, 0) == orc.fft(x, n, 0, inverse=True)


def test_fft_rejects_bad_sizes(bn):
    for n in (0, 3, 6, 1000):
        with pytest.raises(Exception):
            bn.fft(b"\0" * (32 * n), 0)


@pytest.mark.parametrize("g", [1, 2])
def test_msm_golden(bn, orc, g):
    for c in load_golden("msm.json")["g%d" % g]:
        if c["flavour"] == "accumulate_into_3G":
            continue
        s, p = B64(c["scalars"]), B64(c["points"])
        out = bn.g1_multiexp(s, p) if g == 1 else bn.g2_multiexp(s, p)
        want = H(c["multiexp_affine"])
        # affine-normalised Jacobian-Montgomery triple is unique
        assert out == want, (g, c["n"], c["flavour"])


def _points(orc, g, ks):
    gen = H(load_golden("groups.json")["g%d" % g]["gen"])
    sz = 64 if g == 1 else 128
    return b"".join(orc.g_affine(g, orc.g_times_scalar(g, gen, k.to_bytes(32, "little")))[:sz] for k in ks)


@pytest.mark.parametrize("g,n", [(1, 700), (2, 200)])
def test_msm_vs_oracle_skewed(bn, orc, g, n):
    # circuit-like scalars: zeros, ones, small values, hot buckets
    rnd = random.Random(100 + g)
    ks = [rnd.randrange(1, orc.R) for _ in range(n)]
    pts = _points(orc, g, ks)
    sc = []
    for i in range(n):
        u = rnd.random()
        if u < 0.07: v = 0
        elif u < 0.40: v = 1            # hot bucket -> hot-task path
        elif u < 0.55: v = rnd.randrange(1 << 32)
        elif u < 0.60: v = (1 << 256) - 1 - rnd.randrange(1 << 20)
        else: v = rnd.randrange(orc.R)
        sc.append(v.to_bytes(32, "little"))
    sc = b"".join(sc)
    out = bn.g1_multiexp(sc, pts) if g == 1 else bn.g2_multiexp(sc, pts)
    want = orc.g_affine(g, orc.multiexp(g, "multiexp2" if g == 1 else "multiexp", sc, pts, n))
    assert out == want


@pytest.mark.parametrize("g,n", [(1, 700), (2, 200)])
def test_msm_hot_bucket_path(bn, orc, tune, g, n):
    """Buckets with very many tasks (one scalar value shared by most pairs: the ones of a boolean-heavy witness)
    have their task list and their partial sums handled by many workgroups; forced here with a tiny threshold."""
    tune(bn.lib, "MSM_HOT_MIN", 2)
    rnd = random.Random(300 + g)
    ks = [rnd.randrange(1, orc.R) for _ in range(n)]
    pts = _points(orc, g, ks)
    sc = b"".join((1 if rnd.random() < 0.7 else rnd.randrange(orc.R)).to_bytes(32, "little") for _ in range(n))
    out = bn.g1_multiexp(sc, pts) if g == 1 else bn.g2_multiexp(sc, pts)
    assert out == orc.g_affine(g, orc.multiexp(g, "multiexp2" if g == 1 else "multiexp", sc, pts, n))


@pytest.mark.parametrize("g,n,c,hot", [(1, 2000, 9, False), (1, 1500, 10, True), (2, 700, 9, False), (2, 500, 9, True),
                                       (1, 1100, 17, True), (2, 150, 17, False)])
def test_msm_window_widths_hot_and_multi_task_buckets_and_window_shards(bn, orc, tune, g, n, c, hot):
    """A stand-alone MSM at forced window widths, with buckets cut into several tasks and hot buckets (the three roles of the one
    combine launch: a lane, a wavefront, slices + the wavefront that completes a bucket's last slice), zero scalars, raw scalars
    >= r; and the same sum as three window shards combined on the host.  c = 17: plans of <= 16 windows, whose grouping pass takes the
    digits once (presort_scatter_once); c = 9, 10: the general one."""
    tune(bn.lib, "MSM_C", c)
    if hot:
        tune(bn.lib, "MSM_HOT_MIN", 2)
        tune(bn.lib, "MSM_LMAX", 4)
    rnd = random.Random(500 + g + c)
    ks = [rnd.randrange(1, orc.R) for _ in range(n)]
    pts = _points(orc, g, ks)
    sc = b"".join(((1 if rnd.random() < 0.3 else rnd.randrange(1 << 256)) if i % 7 else 0).to_bytes(32, "little") for i in range(n))
    f = bn.g1_multiexp if g == 1 else bn.g2_multiexp
    out = f(sc, pts)
    assert out == orc.g_affine(g, orc.multiexp(g, "multiexp2" if g == 1 else "multiexp", sc, pts, n))
    assert f(sc, pts) == out                                                      # (the hot-bucket completion counters are left at zero)
    parts = b"".join(f(sc, pts, shard=(rank, 3)) for rank in range(3))            # window shards
    assert (bn.g1_sum if g == 1 else bn.g2_sum)(parts) == out



def test_msm_same_point_many_times(bn, orc):
    # every pair identical: exercises the doubling branch of the mixed add and hot buckets
    n = 257
    pts = _points(orc, 1, [5]) * n
    sc = (3).to_bytes(32, "little") * n
    out = bn.g1_multiexp(sc, pts)
    want = orc.g_affine(1, orc.multiexp(1, "multiexp2", sc, pts, n))
    assert out == want


@pytest.mark.parametrize("g", [1, 2])
def test_msm_corner_case_buckets(bn, orc, g):
    from primitives_common import check_msm_corner_case_buckets
    check_msm_corner_case_buckets(bn, orc, g)


def test_calc_h_golden(bn):
    for c in load_golden("calch.json"):
        h = bn.calcH(B64(c["signals"]), B64(c["polsA"]), B64(c["polsB"]), c["nSignals"], c["domain"])
        assert h == B64(c["h"]), (c["nSignals"], c["domain"])


def test_calc_h_fused_passes_vs_oracle(bn, orc):
    """CALC_H with the pointwise products formed by the first pass of the following inverse transform and h stored by the
    last pass of the final one (calch.hip / ntt.hip): two-pass size (per-pass folding tables) against the oracle's
    step-by-step reference sequence (src/bn128.js:139-164)."""
    import struct
    rnd = random.Random(4096)
    nS, dom = 700, 4096
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


def test_new_entry_points_reject_bad_arguments(bn):
    """Round-2 entry points return status codes, never crash: shard out of range, unknown partial flags, layouts that do not
    match the transform size, short witnesses."""
    import ctypes as C
    c = bn.lib.c
    buf = (C.c_uint8 * (64 * 32))()
    out = (C.c_uint8 * 576)()
    assert c.wsnark_g1_msm_windows(buf, buf, 4, 2, 2, out) == 4                 # rank >= world: WSNARK_ERR_ARG
    assert c.wsnark_g1_msm_windows(buf, buf, 4, 0, 0, out) == 4
    assert c.wsnark_fr_dist_scale_dev(buf, 1, 4, 8, 0, 3, 6, 0, 0, None) == 0      # rows 4 of n1 = 8, cols n2 = 8: fine
    assert c.wsnark_fr_dist_scale_dev(buf, 2, 4, 8, 0, 3, 6, 0, 0, None) == 0      # two stacked blocks (2 x 4 x 8 = 64 elements)
    assert c.wsnark_fr_dist_scale_dev(buf, 1, 4, 4, 0, 3, 6, 0, 0, None) == 1      # cols != 2^(log_n - log_n1): WSNARK_ERR_SIZE
    assert c.wsnark_fr_dist_scale_dev(buf, 1, 4, 8, 6, 3, 6, 0, 0, None) == 1      # row0 + rows > n1
    assert c.wsnark_fr_dist_scale_dev(buf, 1, 4, 8, 0, 3, 6, 7, 0, None) == 1      # unknown mode
    assert c.wsnark_fr_ntt_batch_dev(buf, 8, 0, 0, None) == 0                   # count 0: nothing to do
    assert c.wsnark_fr_ntt_batch_dev(buf, 6, 2, 0, None) == 1                   # not a power of two
    assert c.wsnark_fr_dist_combine_dev(buf, buf, buf, 4, 4, 0, 3, 6, None) == 1
    from conftest import GOLDEN
    import os
    key = bn.load_key(open(os.path.join(GOLDEN, "keys", "t3.pkey.bin"), "rb").read())
    wit = open(os.path.join(GOLDEN, "keys", "t3.witness.bin"), "rb").read()
    assert c.wsnark_groth16_prove_partial(key._h, wit, len(wit), 0, 1, 2, out) == 4        # unknown flag bit
    assert c.wsnark_groth16_prove_partial(key._h, wit, 32, 0, 1, 0, out) == 1              # witness too short
    a = (C.c_uint8 * (key.domain * 32))()
    assert c.wsnark_pkey_eval_ab_dev(key._h, wit, 32, a, a, None) == 1
    r = (C.c_uint8 * 32)()
    assert c.wsnark_groth16_verify(None, 0, None, 0, out, C.byref(C.c_int())) == 4


@pytest.mark.parametrize("table_c", [0, 17])
def test_resident_points_give_the_reference_sums(bn, orc, tune, table_c):
    """wsnark_points_load / wsnark_points_msm: a point set resident as fixed-base window tables (no reference counterpart; the layout
    of a resident key's sections) must give the REFERENCE's g1m_multiexp2 / g2m_multiexp results on the golden cases -- x = 0 points,
    duplicates, P / -P pairs, zero scalars, raw scalars up to 2^256 - 1 -- and the oracle's on seeded sets; wrong sizes are errors."""
    import base64
    from wasmsnark_amd import WsnarkError
    if table_c:         # 15 rows: the one-extraction grouping pass on a table plan
        tune(bn.lib, "TABLE_C", table_c)
    for g in (1, 2):
        for k, c in enumerate(load_golden("msm.json")["g%d" % g]):
            if c["flavour"] == "accumulate_into_3G" or c["n"] == 0 or (table_c and (g == 2 or k > 2)):
                continue
            sc, pts = base64.b64decode(c["scalars"]), base64.b64decode(c["points"])
            h = bn.load_points(g, pts)
            assert h.n == c["n"] and h.table["rows"] == -(-255 // h.table["c"])
            assert h.multiexp(sc) == bytes.fromhex(c["multiexp_affine"]), (g, c["n"], c["flavour"])
            assert h.multiexp(sc) == bytes.fromhex(c["multiexp_affine"])           # (again: the handle is reusable)
            h.free()
    rnd = random.Random(4242)
    for g, n in ((1, 1300), (2, 300)) if not table_c else ((1, 700), (2, 60)):
        pts = _points(orc, g, [rnd.randrange(1, orc.R) for _ in range(n)])
        h = bn.load_points(g, pts)
        for _ in range(2):
            sc = b"".join((rnd.randrange(1 << 256) if i % 5 else i % 3).to_bytes(32, "little") for i in range(n))
            assert h.multiexp(sc) == orc.g_affine(g, orc.multiexp(g, "multiexp2" if g == 1 else "multiexp", sc, pts, n))
        with pytest.raises(WsnarkError):
            h.multiexp(sc[:-32])                                                   # one scalar per point
        h.free()
