"""Synthetic Groth16 circuits, proving keys and witnesses in the reference's binary formats.

The reference ships no proving key (test/data/proving_key.bin is missing), so
end-to-end proving runs on synthetic keys built here from KNOWN toxic waste (tau, alpha, beta,
gamma, delta).  Formats follow /root/reference tools/buildpkey.js:124-240 (proving_key.bin) and
tools/buildwitness.js:36-69 (witness.bin); key semantics are snarkjs "groth" (SURVEY.md section 8
row a22): A[s] = a_s(tau) G1, B1/B2[s] = b_s(tau) G, C[s] = ((beta a_s + alpha b_s + c_s)(tau)/delta) G1
for s > nPublic, hExps[i] = (tau^i Z(tau)/delta) G1, pols in the Lagrange basis over w_n.

Because the toxic waste is known, the expected proof for given (r, s) can be written down in the
exponent (expected_proof_scalars), which gives a size-independent end-to-end check that needs no
pairing: the GPU proof must equal (a G1, b G2, c G1).

Group scalar multiplications are delegated to a `mul_base(g, scalars_bytes) -> affine bytes`
callable (the GPU library at full size; tests may pass the CPU oracle for tiny sizes).
"""
import random
import struct

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
W28 = 0x2a3c09f0a58a7e8500e0a7eb8ef62abc402d111e41112ed49bd61b6e725b19f0  # 5^((r-1)/2^28), src/build_fft.js:29-47
MONT = 1 << 256


def root_of_unity(bits):
    w = W28
    for _ in range(28 - bits):
        w = w * w % R
    return w


def le32(v):
    return int(v).to_bytes(32, "little")


class Circuit:
    """R1CS in column form: cols[M][s] = {constraint: coef} for M in A,B,C."""

    def __init__(self, n_vars, n_public, domain, A, B, Cm, witness):
        self.n_vars, self.n_public, self.domain = n_vars, n_public, domain
        self.A, self.B, self.C = A, B, Cm
        self.witness = witness


def make_circuit(log_domain, n_public=2, seed=1, nnz_extra=0.5, style="columns"):
    """Multiplication-chain circuit with domainSize = 2^log_domain exactly:
    nConstraints = domain - nPublic - 1 'real' rows, then the nPublic+1 input-binding rows
    (w_i * 0 = 0) old snarkjs appends.  Every constraint defines one
    fresh variable: w[out] = (sum A w)(sum B w), so the witness is computed on the fly.

    style="columns" (default; the shape SURVEY.md section 8d C4 specifies): every variable occurs in 1-3 rows
    of A and in 1-3 rows of B (rows that come after its own definition; a row left empty gets one fill-in term),
    so every key point of A, B1, B2 is a real point -- except for the very last variable, which no later row can use.
    style="rows" (round 1): 1-2 terms per ROW drawn among the earlier variables; ~40 % of the variables then
    never occur in A (resp. B) and their key points are infinity -- the sparse case the plan variants exploit."""
    rnd = random.Random(seed)
    domain = 1 << log_domain
    n_cons = domain - n_public - 1
    if n_cons < 1:
        raise ValueError("domain too small for nPublic")
    n_free = n_public + 2                      # public inputs + two private seeds
    n_vars = 1 + n_free + n_cons
    w = [0] * n_vars
    w[0] = 1
    for i in range(1, 1 + n_free):
        w[i] = rnd.randrange(1, R) if i % 3 else rnd.randrange(1, 1 << 32)
    A = [dict() for _ in range(n_vars)]
    B = [dict() for _ in range(n_vars)]
    Cm = [dict() for _ in range(n_vars)]
    coef_of = lambda: rnd.randrange(1, R) if rnd.random() < 0.5 else rnd.randrange(1, 8)
    if style == "columns":
        rows = {id(A): [[] for _ in range(n_cons)], id(B): [[] for _ in range(n_cons)]}
        for M in (A, B):
            rw = rows[id(M)]
            for s in range(n_vars):
                c_min = max(0, s - n_free)          # first row whose output variable comes after s
                if c_min >= n_cons:
                    continue
                for _ in range(rnd.randrange(1, 4)):
                    c = rnd.randrange(c_min, n_cons)
                    if c not in M[s]:
                        M[s][c] = cf = coef_of()
                        rw[c].append((s, cf))
            for c in range(n_cons):
                if not rw[c]:
                    s = rnd.randrange(0, 1 + n_free + c)
                    M[s][c] = cf = coef_of()
                    rw[c].append((s, cf))
        rA, rB = rows[id(A)], rows[id(B)]
        for c in range(n_cons):
            out = 1 + n_free + c
            lhs = sum(cf * w[s] for s, cf in rA[c]) % R
            rhs = sum(cf * w[s] for s, cf in rB[c]) % R
            Cm[out][c] = 1
            w[out] = lhs * rhs % R
    elif style == "rows":
        for c in range(n_cons):
            out = 1 + n_free + c
            lhs = rhs = 0
            for M, acc in ((A, 0), (B, 1)):
                k = 1 + (1 if rnd.random() < nnz_extra else 0)
                tot = 0
                used = set()
                for _ in range(k):
                    s = rnd.randrange(0, out)
                    if s in used:
                        continue
                    used.add(s)
                    coef = coef_of()
                    M[s][c] = coef
                    tot = (tot + coef * w[s]) % R
                if acc == 0:
                    lhs = tot
                else:
                    rhs = tot
            Cm[out][c] = 1
            w[out] = lhs * rhs % R
    else:
        raise ValueError("style must be 'columns' or 'rows'")
    for i in range(n_public + 1):              # input-binding rows: A = w_i, B = 0, C = 0
        A[i][n_cons + i] = 1
    return Circuit(n_vars, n_public, domain, A, B, Cm, w)


def lagrange_at(tau, log_domain):
    """L_c(tau) for the domain {w^c}: (tau^n - 1)/n * w^c/(tau - w^c), by batch inversion."""
    n = 1 << log_domain
    w = root_of_unity(log_domain)
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * w % R
    den = [(tau - x) % R for x in pw]
    pref = [1] * (n + 1)
    for i in range(n):
        pref[i + 1] = pref[i] * den[i] % R
    inv = pow(pref[n], R - 2, R)
    z = (pow(tau, n, R) - 1) % R
    zn = z * pow(n, R - 2, R) % R
    L = [0] * n
    for i in range(n - 1, -1, -1):
        di = inv * pref[i] % R
        inv = inv * den[i] % R
        L[i] = zn * pw[i] % R * di % R
    return L, z


class Setup:
    pass


def setup(circ, seed=7):
    """Toxic waste + per-signal scalars a_s, b_s, c_s at tau."""
    rnd = random.Random(seed)
    S = Setup()
    S.tau, S.alpha, S.beta, S.gamma, S.delta = (rnd.randrange(2, R) for _ in range(5))
    log_domain = circ.domain.bit_length() - 1
    L, S.z = lagrange_at(S.tau, log_domain)
    ev = lambda col: sum(coef * L[c] for c, coef in col.items()) % R
    S.a = [ev(c) for c in circ.A]
    S.b = [ev(c) for c in circ.B]
    S.c = [ev(c) for c in circ.C]
    return S


def _pol_blob(cols):
    # tools/buildpkey.js:79-89 writeTransformedPolynomial: u32 count, then (u32 idx, 32 B coef Montgomery)
    out = bytearray()
    for col in cols:
        out += struct.pack("<I", len(col))
        for idx, coef in col.items():
            out += struct.pack("<I", idx) + le32(coef * MONT % R)
    return bytes(out)


def build_sections(circ, S, mul_base):
    """The key as separate sections (the input of wsnark_pkey_load_sections / Bn128.load_key(sections=...)):
    what proving_key.bin holds, without its u32 offsets -- keys beyond 4 GiB (2^23 constraints and up) only
    exist in this form.  Returns (sections dict, (IC points, beta2/delta2/gamma2 bytes) for the verification key)."""
    nv, npub, dom = circ.n_vars, circ.n_public, circ.domain
    dinv, ginv = pow(S.delta, R - 2, R), pow(S.gamma, R - 2, R)
    kc = [(S.beta * S.a[s] + S.alpha * S.b[s] + S.c[s]) % R for s in range(nv)]
    hs, t = [], S.z * dinv % R
    for _ in range(dom):
        hs.append(t)
        t = t * S.tau % R
    cat = lambda xs: b"".join(le32(x) for x in xs)
    g1_scalars = ([S.alpha, S.beta, S.delta] + S.a + S.b + [kc[s] * dinv % R for s in range(npub + 1, nv)] + hs
                  + [kc[s] * ginv % R for s in range(npub + 1)])
    g1 = mul_base(1, cat(g1_scalars))
    g2 = mul_base(2, cat([S.beta, S.delta, S.gamma] + S.b))
    P1 = lambda i: g1[64 * i:64 * (i + 1)]
    P2 = lambda i: g2[128 * i:128 * (i + 1)]
    o = 3
    ptsA = g1[64 * o:64 * (o + nv)]; o += nv
    ptsB1 = g1[64 * o:64 * (o + nv)]; o += nv
    nC = nv - npub - 1
    ptsC = g1[64 * o:64 * (o + nC)]; o += nC
    ptsH = g1[64 * o:64 * (o + dom)]; o += dom
    ic = [P1(o + i) for i in range(npub + 1)]
    sec = {"n_vars": nv, "n_public": npub, "domain": dom, "alfa1": P1(0), "beta1": P1(1), "delta1": P1(2),
           "beta2": P2(0), "delta2": P2(1), "polsA": _pol_blob(circ.A), "polsB": _pol_blob(circ.B),
           "pointsA": ptsA, "pointsB1": ptsB1, "pointsB2": g2[128 * 3:128 * (3 + nv)], "pointsC": ptsC, "pointsH": ptsH}
    return sec, (ic, P2(2))


def build_key(circ, S, mul_base):
    """Returns (proving_key.bin bytes, verification key dict in the reference's JSON shape)."""
    npub = circ.n_public
    sec, (ic, gamma2) = build_sections(circ, S, mul_base)
    P1 = lambda i: (sec["alfa1"], sec["beta1"], sec["delta1"])[i]
    P2 = lambda i: (sec["beta2"], sec["delta2"], gamma2)[i]
    # tools/buildpkey.js:124-186 layout
    fixed = sec["alfa1"] + sec["beta1"] + sec["delta1"] + sec["beta2"] + sec["delta2"]
    parts = [sec["polsA"], sec["polsB"], sec["pointsA"], sec["pointsB1"], sec["pointsB2"], sec["pointsC"], sec["pointsH"]]
    offs, o = [], 40 + len(fixed)
    for part in parts:
        offs.append(o)
        o += len(part)
    header = struct.pack("<10I", sec["n_vars"], npub, sec["domain"], *offs)
    pkey = header + fixed + b"".join(parts)

    def dec1(p):  # Montgomery affine bytes -> decimal strings (x, y, 1)
        rinv = pow(MONT, Q - 2, Q)
        x = int.from_bytes(p[:32], "little") * rinv % Q
        y = int.from_bytes(p[32:64], "little") * rinv % Q
        return [str(x), str(y), "1"]

    def dec2(p):
        rinv = pow(MONT, Q - 2, Q)
        v = [str(int.from_bytes(p[i:i + 32], "little") * rinv % Q) for i in range(0, 128, 32)]
        return [[v[0], v[1]], [v[2], v[3]], ["1", "0"]]

    vk = {"protocol": "groth", "nPublic": npub, "vk_alfa_1": dec1(P1(0)), "vk_beta_2": dec2(P2(0)),
          "vk_gamma_2": dec2(P2(2)), "vk_delta_2": dec2(P2(1)), "IC": [dec1(p) for p in ic]}
    return pkey, vk


def witness_bin(circ):
    return b"".join(le32(x) for x in circ.witness)   # tools/buildwitness.js:36-41: plain LE


def public_signals(circ):
    return [str(x) for x in circ.witness[1:circ.n_public + 1]]


def expected_proof_scalars(circ, S, r32, s32):
    """Discrete logs (a, b, c) of the proof elements w.r.t. G1/G2/G1 for blinding bytes r32, s32."""
    r = int.from_bytes(r32, "little") % R
    s = int.from_bytes(s32, "little") % R
    w = circ.witness
    Aw = sum(x * y for x, y in zip(w, S.a)) % R
    Bw = sum(x * y for x, y in zip(w, S.b)) % R
    Cw = sum(x * y for x, y in zip(w, S.c)) % R
    dinv = pow(S.delta, R - 2, R)
    h_tau_z = (Aw * Bw - Cw) % R                   # h(tau) Z(tau) = A(tau)B(tau) - C(tau)
    a = (S.alpha + Aw + r * S.delta) % R
    b = (S.beta + Bw + s * S.delta) % R
    priv = sum(w[i] * ((S.beta * S.a[i] + S.alpha * S.b[i] + S.c[i]) % R) for i in range(circ.n_public + 1, circ.n_vars)) % R
    c = ((priv + h_tau_z) * dinv + s * a + r * b - r * s % R * S.delta) % R
    return a, b, c


def expected_proof(circ, S, r32, s32, mul_base):
    """The proof object (decimal strings, reference shape) computed from the toxic waste."""
    a, b, c = expected_proof_scalars(circ, S, r32, s32)
    g1 = mul_base(1, le32(a) + le32(c))
    g2 = mul_base(2, le32(b))
    rinv = pow(MONT, Q - 2, Q)
    dec = lambda bs: str(int.from_bytes(bs, "little") * rinv % Q)
    def p1(p):
        return ["0", "1", "0"] if p[:32] == b"\0" * 32 else [dec(p[:32]), dec(p[32:64]), "1"]
    pa, pc = p1(g1[:64]), p1(g1[64:128])
    pb = [[dec(g2[0:32]), dec(g2[32:64])], [dec(g2[64:96]), dec(g2[96:128])], ["1", "0"]]
    return {"pi_a": pa, "pi_b": pb, "pi_c": pc}


class _SynthInfo(__import__("ctypes").Structure):   # wsnark_synth_info_t
    import ctypes as _C
    _fields_ = [("n_vars", _C.c_uint32), ("n_public", _C.c_uint32), ("domain", _C.c_uint32),
                ("nnz_a", _C.c_uint64), ("nnz_b", _C.c_uint64), ("absent_a", _C.c_uint64), ("absent_b", _C.c_uint64),
                ("pols_a_len", _C.c_uint64), ("pols_b_len", _C.c_uint64), ("n_g1_scalars", _C.c_uint64), ("n_g2_scalars", _C.c_uint64)]


class NativeCircuit:
    """The same construction as make_circuit + setup + build_sections + expected_proof, done by the library's host-side
    generator (csrc/synth.hip, wsnark_synth_*): seconds instead of minutes at 2^22-2^24, which is what lets BASELINE
    config 5's size run under the GPU tests.  `lib` is the loaded C ABI (wasmsnark_amd._lib.Lib; the group scalar
    multiplications go to its wsnark_g{1,2}_mul_base_batch).  Different random stream than the Python generator: the
    circuits are not the same instances, only the same family (tests/test_synth_native.py pins this generator against
    the oracle's prover and the native verifier)."""

    def __init__(self, lib, log_domain, n_public=2, seed=1, setup_seed=None, style="columns"):
        import ctypes as C
        self._lib, self._h = lib, C.c_void_p()
        st = {"columns": 0, "rows": 1, "boolean": 2}[style]      # "boolean": bit decompositions, 87.5 % of the witness 0 / 1 (csrc/synth.hip)
        lib.check(lib.c.wsnark_synth_new(log_domain, n_public, seed, seed + 1 if setup_seed is None else setup_seed, st, C.byref(self._h)))
        inf = _SynthInfo()
        lib.check(lib.c.wsnark_synth_info(self._h, C.byref(inf)))
        self.info = inf
        self.n_vars, self.n_public, self.domain = inf.n_vars, inf.n_public, inf.domain
        self.nnz = inf.nnz_a + inf.nnz_b
        self.absent = (inf.absent_a, inf.absent_b)
        self.style = style
        self._wit = None

    def free(self):
        if self._h:
            self._lib.c.wsnark_synth_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass

    @staticmethod
    def _cbuf(ba, off=0, n=None):
        import ctypes as C
        n = len(ba) - off if n is None else n
        return (C.c_uint8 * max(n, 1)).from_buffer(ba, off) if n else None

    def witness_bin(self):
        if self._wit is None:
            out = bytearray(self.n_vars * 32)
            self._lib.check(self._lib.c.wsnark_synth_witness(self._h, self._cbuf(out)))
            self._wit = bytes(out)
        return self._wit

    def public_signals(self):
        w = self.witness_bin()
        return [str(int.from_bytes(w[32 * i:32 * i + 32], "little")) for i in range(1, self.n_public + 1)]

    def _mul(self, g, scalars, off, n, base=None):
        """n points k_i * generator for the scalars at byte offset `off`, as a bytearray (no intermediate copies)."""
        from .bn128 import G1_GEN, G2_GEN
        sz = 64 if g == 1 else 128
        out = bytearray(n * sz)
        if n:
            fn = self._lib.c.wsnark_g1_mul_base_batch if g == 1 else self._lib.c.wsnark_g2_mul_base_batch
            self._lib.check(fn(G1_GEN if g == 1 else G2_GEN, self._cbuf(scalars, off, n * 32), n, self._cbuf(out)))
        return out

    def build_sections(self, mul_base=None):
        """(sections dict for Bn128.load_key(sections=...), (IC points, gamma2 bytes)); `mul_base` is ignored (the
        library's own fixed-base kernel is used), the parameter only mirrors synth.build_sections."""
        inf, nv, npub, dom = self.info, self.n_vars, self.n_public, self.domain
        s1 = bytearray(inf.n_g1_scalars * 32)
        s2 = bytearray(inf.n_g2_scalars * 32)
        self._lib.check(self._lib.c.wsnark_synth_key_scalars(self._h, 1, self._cbuf(s1)))
        self._lib.check(self._lib.c.wsnark_synth_key_scalars(self._h, 2, self._cbuf(s2)))
        nC = nv - npub - 1
        o = 0
        fixed1 = self._mul(1, s1, 0, 3); o += 3
        ptsA = self._mul(1, s1, o * 32, nv); o += nv
        ptsB1 = self._mul(1, s1, o * 32, nv); o += nv
        ptsC = self._mul(1, s1, o * 32, nC); o += nC
        ptsH = self._mul(1, s1, o * 32, dom); o += dom
        ic = self._mul(1, s1, o * 32, npub + 1)
        fixed2 = self._mul(2, s2, 0, 3)
        ptsB2 = self._mul(2, s2, 96, nv)
        polsA, polsB = bytearray(inf.pols_a_len), bytearray(inf.pols_b_len)
        self._lib.check(self._lib.c.wsnark_synth_pols(self._h, 0, self._cbuf(polsA), len(polsA)))
        self._lib.check(self._lib.c.wsnark_synth_pols(self._h, 1, self._cbuf(polsB), len(polsB)))
        sec = {"n_vars": nv, "n_public": npub, "domain": dom,
               "alfa1": bytes(fixed1[0:64]), "beta1": bytes(fixed1[64:128]), "delta1": bytes(fixed1[128:192]),
               "beta2": bytes(fixed2[0:128]), "delta2": bytes(fixed2[128:256]), "polsA": polsA, "polsB": polsB,
               "pointsA": ptsA, "pointsB1": ptsB1, "pointsB2": ptsB2, "pointsC": ptsC, "pointsH": ptsH}
        return sec, ([bytes(ic[64 * i:64 * i + 64]) for i in range(npub + 1)], bytes(fixed2[256:384]))

    def build_key(self, mul_base=None):
        """(proving_key.bin bytes, verification key dict) -- only below the 4 GiB of the file format's u32 offsets."""
        sec, (ic, gamma2) = self.build_sections()
        return sections_to_pkey(sec), vk_from_points(self.n_public, sec, ic, gamma2)

    def expected_scalars(self, r32, s32):
        out = bytearray(96)
        self._lib.check(self._lib.c.wsnark_synth_expected(self._h, bytes(r32), bytes(s32), self._cbuf(out)))
        return tuple(int.from_bytes(out[32 * i:32 * i + 32], "little") for i in range(3))

    def expected_proof(self, r32, s32, mul_base=None):
        a, b, c = self.expected_scalars(r32, s32)
        sc = bytearray(le32(a) + le32(c))
        return proof_from_points(bytes(self._mul(1, sc, 0, 2)), bytes(self._mul(2, bytearray(le32(b)), 0, 1)))


def sections_to_pkey(sec):
    """tools/buildpkey.js:124-186 layout from the separate sections."""
    fixed = bytes(sec["alfa1"]) + bytes(sec["beta1"]) + bytes(sec["delta1"]) + bytes(sec["beta2"]) + bytes(sec["delta2"])
    parts = [sec["polsA"], sec["polsB"], sec["pointsA"], sec["pointsB1"], sec["pointsB2"], sec["pointsC"], sec["pointsH"]]
    offs, o = [], 40 + len(fixed)
    for part in parts:
        offs.append(o)
        o += len(part)
    if o >= 1 << 32:
        raise ValueError("key beyond the 4 GiB of proving_key.bin's u32 offsets: use the sections container")
    return struct.pack("<10I", sec["n_vars"], sec["n_public"], sec["domain"], *offs) + fixed + b"".join(bytes(p) for p in parts)


def _dec_q(bs):
    return str(int.from_bytes(bs, "little") * pow(MONT, Q - 2, Q) % Q)


def vk_from_points(npub, sec, ic, gamma2):
    dec1 = lambda p: [_dec_q(p[:32]), _dec_q(p[32:64]), "1"]
    dec2 = lambda p: [[_dec_q(p[0:32]), _dec_q(p[32:64])], [_dec_q(p[64:96]), _dec_q(p[96:128])], ["1", "0"]]
    return {"protocol": "groth", "nPublic": npub, "vk_alfa_1": dec1(sec["alfa1"]), "vk_beta_2": dec2(sec["beta2"]),
            "vk_gamma_2": dec2(gamma2), "vk_delta_2": dec2(sec["delta2"]), "IC": [dec1(p) for p in ic]}


def proof_from_points(g1, g2):
    """{pi_a, pi_c} = the two 64-byte affine Montgomery points of g1, pi_b = the 128-byte point g2, as the proof object."""
    p1 = lambda p: ["0", "1", "0"] if p[:32] == b"\0" * 32 else [_dec_q(p[:32]), _dec_q(p[32:64]), "1"]
    pb = [[_dec_q(g2[0:32]), _dec_q(g2[32:64])], [_dec_q(g2[64:96]), _dec_q(g2[96:128])], ["1", "0"]]
    return {"pi_a": p1(g1[:64]), "pi_b": pb, "pi_c": p1(g1[64:128])}


def pseudo_key(n_vars, n_public, domain, seed, mul_base):
    """A well-formed but NOT circuit-valid proving key with the given header: every point is
    k * G for seeded pseudo-random k, polsA/polsB have one pseudo-random coefficient per signal.
    groth16GenProof is a pure function of (witness, key, r, s), so such a key is enough to compare
    provers bit for bit on a real witness whose proving key is not available (BASELINE config 1:
    the reference's example/bn128/witness.bin)."""
    rnd = random.Random(seed)
    nC = n_vars - n_public - 1
    ks1 = [rnd.randrange(1, R) for _ in range(3 + 2 * n_vars + nC + domain)]
    ks2 = [rnd.randrange(1, R) for _ in range(2 + n_vars)]
    # a few points at infinity, as real keys have for signals absent from B
    for i in range(0, n_vars, 97):
        ks1[3 + n_vars + i] = 0
        ks2[2 + i] = 0
    g1 = mul_base(1, b"".join(le32(k) for k in ks1))
    g2 = mul_base(2, b"".join(le32(k) for k in ks2))

    def pols():
        out = bytearray()
        for _ in range(n_vars):
            out += struct.pack("<I", 1) + struct.pack("<I", rnd.randrange(domain)) + le32(rnd.randrange(R))
        return bytes(out)

    polsA, polsB = pols(), pols()
    fixed = g1[:192] + g2[:256]
    o = 3
    ptsA = g1[64 * o:64 * (o + n_vars)]; o += n_vars
    ptsB1 = g1[64 * o:64 * (o + n_vars)]; o += n_vars
    ptsC = g1[64 * o:64 * (o + nC)]; o += nC
    ptsH = g1[64 * o:64 * (o + domain)]
    ptsB2 = g2[256:256 + 128 * n_vars]
    pPolsA = 40 + len(fixed)
    pPolsB = pPolsA + len(polsA)
    pA = pPolsB + len(polsB)
    pB1 = pA + len(ptsA)
    pB2 = pB1 + len(ptsB1)
    pC = pB2 + len(ptsB2)
    pH = pC + len(ptsC)
    header = struct.pack("<10I", n_vars, n_public, domain, pPolsA, pPolsB, pA, pB1, pB2, pC, pH)
    return header + fixed + polsA + polsB + ptsA + ptsB1 + ptsB2 + ptsC + ptsH
