"""Host-side mirror of the reference's JS API for the prove path, over the C ABI.

Mirrors /root/reference src/bn128.js (class Bn128, `build()`), main_bn128.js
(window.groth16GenProof) and README.md:28-30 (genZKSnarkProof): same names, same argument
meaning (byte buffers in the reference's layouts), same return shapes (Jacobian-Montgomery
byte strings for the multiexps, plain-form h for calcH, an object of decimal strings for
proofs).  The Node.js drop-in (wasmsnark_amd/js) binds the same C ABI through N-API; this
Python mirror exists so the parity tests read like the reference's own tests.
"""
import ctypes as C
import os
import weakref

from . import _lib


_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def _mont(v):
    return ((v << 256) % _Q).to_bytes(32, "little")


# generators (src/bn128/build_bn128.js:59-90), affine Montgomery
G1_GEN = _mont(1) + _mont(2)
G2_GEN = (_mont(10857046999023057135944570762232829481370756359578518086990519993285655852781)
          + _mont(11559732032986387107991004021392285783925812861821192530917403151452391805634)
          + _mont(8495653923123431417604973247489272438418190587263600148770280649306958101930)
          + _mont(4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _buf(b):
    """Private mutable copy (for entry points that work in place)."""
    if isinstance(b, (bytes, bytearray, memoryview)):
        b = bytes(b)
        return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0"), len(b)
    raise TypeError("expected a bytes-like object (the reference takes ArrayBuffers)")


def _ro(b):
    """Read-only input without a copy: the C side borrows the caller's bytes for the duration of the call
    (a 2^20-pair MSM input is 96 MB; copying it in Python cost more than the MSM)."""
    if isinstance(b, bytes):
        return (C.c_char_p(b) if len(b) else C.c_char_p(b"\0")), len(b)
    if isinstance(b, bytearray):
        return ((C.c_uint8 * len(b)).from_buffer(b) if len(b) else C.c_char_p(b"\0")), len(b)
    if isinstance(b, memoryview):
        return _ro(b.obj if isinstance(b.obj, (bytes, bytearray)) and b.nbytes == len(b.obj) else bytes(b))
    raise TypeError("expected a bytes-like object (the reference takes ArrayBuffers)")


class _KeySections(C.Structure):   # wsnark_key_sections_t
    _fields_ = [("n_vars", C.c_uint32), ("n_public", C.c_uint32), ("domain", C.c_uint32),
                ("alfa1", C.c_void_p), ("beta1", C.c_void_p), ("delta1", C.c_void_p),
                ("beta2", C.c_void_p), ("delta2", C.c_void_p),
                ("polsA", C.c_void_p), ("polsA_len", C.c_uint64), ("polsB", C.c_void_p), ("polsB_len", C.c_uint64),
                ("pointsA", C.c_void_p), ("pointsA_len", C.c_uint64), ("pointsB1", C.c_void_p), ("pointsB1_len", C.c_uint64),
                ("pointsB2", C.c_void_p), ("pointsB2_len", C.c_uint64), ("pointsC", C.c_void_p), ("pointsC_len", C.c_uint64),
                ("pointsH", C.c_void_p), ("pointsH_len", C.c_uint64)]


class ProvingKey:
    """Device-resident proving key (wsnark_pkey_load, or wsnark_pkey_load_sections for `sections`)."""

    def __init__(self, lib, data=None, sections=None, shard=None, h_interleave_log=0, wait_tables=True, path=None):
        """path: a key FILE -- proving_key.bin or the WSNARK64 container for keys beyond 4 GiB (wsnark_pkey_load_file: mapped,
        only the share's pages are read); shard / h_interleave_log apply to it as to `sections`.
        shard=(rank, world) with `sections`: only that rank's share of the points becomes resident
        (wsnark_pkey_load_shard; h_interleave_log: the layout of its hExps share, see include/wsnark.h).
        wait_tables: the library builds the fixed-base table rows in the background and serves proofs from the plain sections
        until they are there; this mirror waits for them by default (tests and timing tools want the steady state from the first
        call); wait_tables=False returns as the C call does -- `load_ms` then holds what the caller waited for, and
        `wait_tables()` / `refresh_load_stats()` complete the picture later."""
        self._lib = lib
        self._h = C.c_void_p()
        if shard is not None and sections is None and path is None:
            raise ValueError("a points shard is loaded from sections or from a key file")
        if path is not None:
            rank, world = shard if shard is not None else (0, 1)
            lib.check(lib.c.wsnark_pkey_load_file(os.fsencode(path), rank, world, h_interleave_log, C.byref(self._h)))
        elif sections is not None:
            # dict: n_vars, n_public, domain + byte strings alfa1, beta1, delta1, beta2, delta2, polsA, polsB,
            # pointsA, pointsB1, pointsB2, pointsC, pointsH (64-bit lengths: keys beyond the 4 GiB file format)
            ks = _KeySections(sections["n_vars"], sections["n_public"], sections["domain"])
            keep = []
            for name in ("alfa1", "beta1", "delta1", "beta2", "delta2", "polsA", "polsB", "pointsA", "pointsB1",
                         "pointsB2", "pointsC", "pointsH"):
                b, n = _ro(sections[name])
                keep.append(b)
                setattr(ks, name, C.cast(b, C.c_void_p))
                if name.startswith(("pols", "points")):
                    setattr(ks, name + "_len", n)     # the library checks every section against the header
            if shard is not None:
                lib.check(lib.c.wsnark_pkey_load_shard(C.byref(ks), shard[0], shard[1], h_interleave_log, C.byref(self._h)))
            else:
                lib.check(lib.c.wsnark_pkey_load_sections(C.byref(ks), C.byref(self._h)))
        else:
            b, n = _ro(data)
            lib.check(lib.c.wsnark_pkey_load(b, n, C.byref(self._h)))
        nv, npub, dom = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.check(lib.c.wsnark_pkey_info(self._h, C.byref(nv), C.byref(npub), C.byref(dom)))
        self.n_vars, self.n_public, self.domain = nv.value, npub.value, dom.value
        cw, rw, ch, rh, nb = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        lib.check(lib.c.wsnark_pkey_table_info(self._h, C.byref(cw), C.byref(rw), C.byref(ch), C.byref(rh), C.byref(nb)))
        # how the point sections are resident: fixed-base window tables (rows > 1) or the plain sections
        self.table = {"c_w": cw.value, "rows_w": rw.value, "c_h": ch.value, "rows_h": rh.value, "bytes": nb.value}
        if wait_tables:
            lib.check(lib.c.wsnark_pkey_wait_tables(self._h))
        self.refresh_load_stats()
        rk, wd, hl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lo, nl, nh = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib.check(lib.c.wsnark_pkey_shard_info(self._h, C.byref(rk), C.byref(wd), C.byref(lo), C.byref(nl), C.byref(nh), C.byref(hl)))
        # which share of the points this handle holds (a whole key: rank 0 of 1, all nVars signals, all hExps)
        self.shard = {"rank": rk.value, "world": wd.value, "first_signal": lo.value, "n_signals": nl.value, "n_hexps": nh.value,
                      "h_interleave_log": hl.value}

    def wait_tables(self):
        """Block until the background build of the table rows is over (wsnark_pkey_wait_tables)."""
        self._lib.check(self._lib.c.wsnark_pkey_wait_tables(self._h))
        self.refresh_load_stats()

    def refresh_load_stats(self):
        ms = (C.c_double * 5)()
        self._lib.check(self._lib.c.wsnark_pkey_load_stats(self._h, ms))
        # total = what the load call took; table_build = the background build's own duration (0 while it is still running)
        self.load_ms = {"pols_to_csr": ms[0], "points_h2d": ms[1], "masks_convert": ms[2], "table_build": ms[3], "total": ms[4]}

    def free(self):
        if self._h:
            self._lib.c.wsnark_pkey_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ResidentPoints:
    """A point set kept on the device as fixed-base window tables (wsnark_points_load): sums over the same bases without the
    points' H2D copy, the per-window plans and the host's doubling chain.  g: 1 (G1) or 2 (G2)."""

    def __init__(self, lib, g, points):
        self._lib, self.g = lib, g
        b, nbytes = _ro(points)
        sz = 128 if g == 2 else 64
        if nbytes % sz:
            raise ValueError("points: not a whole number of %d-byte points" % sz)
        self.n = nbytes // sz
        self._h = C.c_void_p()
        lib.check(lib.c.wsnark_points_load(g, b, self.n, C.byref(self._h)))
        gg, n, c, rows, nb = C.c_int(), C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        lib.check(lib.c.wsnark_points_info(self._h, C.byref(gg), C.byref(n), C.byref(c), C.byref(rows), C.byref(nb)))
        self.table = {"c": c.value, "rows": rows.value, "bytes": nb.value}

    def multiexp(self, scalars):
        b, n = _ro(scalars)
        out = (C.c_uint8 * (192 if self.g == 2 else 96))()
        self._lib.check(self._lib.c.wsnark_points_msm(self._h, b, n // 32, out))
        return bytes(out)

    def multiexp_dev(self, d_scalars, n, stream=None):
        out = (C.c_uint8 * (192 if self.g == 2 else 96))()
        self._lib.check(self._lib.c.wsnark_points_msm_dev(self._h, d_scalars, n, out, stream))
        return bytes(out)

    def free(self):
        if self._h:
            self._lib.c.wsnark_points_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _key_sections(sections):
    """dict of byte strings -> (the C struct wsnark_key_sections_t, the buffers it points into)"""
    ks = _KeySections(sections["n_vars"], sections["n_public"], sections["domain"])
    keep = []
    for name in ("alfa1", "beta1", "delta1", "beta2", "delta2", "polsA", "polsB", "pointsA", "pointsB1",
                 "pointsB2", "pointsC", "pointsH"):
        b, n = _ro(sections[name])
        keep.append(b)
        setattr(ks, name, C.cast(b, C.c_void_p))
        if name.startswith(("pols", "points")):
            setattr(ks, name + "_len", n)
    return ks, keep


class GroupKey:
    """One points shard of a proving key per device of a Group (wsnark_group_pkey_load[_sections])."""

    def __init__(self, group, data=None, sections=None, wait_tables=True, path=None):
        self._group, self._lib = group, group._lib
        lib = self._lib
        self._h = C.c_void_p()
        if path is not None:
            lib.check(lib.c.wsnark_group_pkey_load_file(group._h, os.fsencode(path), C.byref(self._h)))
        elif sections is not None:
            ks, keep = _key_sections(sections)
            lib.check(lib.c.wsnark_group_pkey_load_sections(group._h, C.byref(ks), C.byref(self._h)))
        else:
            b, n = _ro(data)
            lib.check(lib.c.wsnark_group_pkey_load(group._h, b, n, C.byref(self._h)))
        nv, npub, dom, world, dist = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int()
        lib.check(lib.c.wsnark_group_pkey_info(self._h, C.byref(nv), C.byref(npub), C.byref(dom), C.byref(world), C.byref(dist)))
        self.n_vars, self.n_public, self.domain, self.world = nv.value, npub.value, dom.value, world.value
        self.distributed_calc_h = bool(dist.value)      # CALC_H on the four-step transform (else complete on every device)
        group._keys.add(self)       # wsnark_group_free deletes every key of the group: terminate() forgets their handles first
        if wait_tables:
            lib.check(lib.c.wsnark_group_pkey_wait_tables(self._h))

    def free(self):
        """Frees the shard on every device.  A no-op once the group is gone: wsnark_group_free has freed the group's keys with it
        (include/wsnark.h: "group_free invalidates every key handle of the group")."""
        if self._h and self._group._h:
            self._lib.c.wsnark_group_pkey_free(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Group:
    """Several GPUs in ONE process (wsnark_group_*, csrc/group.hip): the reference's worker pool (src/bn128.js:173-265, 353-415) with
    GPUs as the workers and the transport inside the library.  devices: HIP ordinals (an ordinal may repeat: two contexts on one GPU)."""

    def __init__(self, lib=None, devices=(0,)):
        self._lib = lib or _lib.load()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        self._keys = weakref.WeakSet()
        self._lib.check(self._lib.c.wsnark_group_create(arr, len(self.devices), C.byref(self._h)))

    def load_key(self, pkey=None, sections=None, wait_tables=True, path=None):
        return GroupKey(self, data=pkey, sections=sections, wait_tables=wait_tables, path=path)

    def groth16GenProof(self, signals, key, r=None, s=None):
        """src/bn128.js:580-720 over the group: `signals` the witness bytes (host memory), `key` a GroupKey (or the key's bytes)."""
        own = None
        if not isinstance(key, GroupKey):
            own = key = self.load_key(key)
        try:
            b, n = _ro(signals)
            out = (C.c_uint8 * 384)()
            rb = _ro(r)[0] if r is not None else None
            sb = _ro(s)[0] if s is not None else None
            self._lib.check(self._lib.c.wsnark_group_prove(key._h, b, n, rb, sb, out))
            return proof_from_bytes(bytes(out))
        finally:
            if own is not None:
                own.free()

    def last_blinding(self):
        """(r, s) of the group's last proof -- the reference's this._pr / this._ps (src/bn128.js:662-664)."""
        r, s = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
        self._lib.check(self._lib.c.wsnark_group_last_blinding(self._h, r, s))
        return bytes(r), bytes(s)

    def _multiexp(self, g, scalars, points):
        sb, sn = _ro(scalars)
        pb, pn = _ro(points)
        n = sn // 32
        if pn != n * (128 if g else 64):
            raise ValueError("points length does not match scalars")
        out = (C.c_uint8 * (192 if g else 96))()
        fn = self._lib.c.wsnark_group_g2_msm if g else self._lib.c.wsnark_group_g1_msm
        self._lib.check(fn(self._h, sb, pb, n, out))
        return bytes(out)

    def g1_multiexp(self, scalars, points):      # src/bn128.js:353-383
        return self._multiexp(0, scalars, points)

    def g2_multiexp(self, scalars, points):      # src/bn128.js:385-415
        return self._multiexp(1, scalars, points)

    def terminate(self):                         # src/bn128.js:562-566
        if self._h:
            for k in list(self._keys):           # the library frees the group's keys with the group: their handles die here
                k._h = C.c_void_p()
            self._lib.c.wsnark_group_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.terminate()
        except Exception:
            pass


class Bn128:
    """`await buildBn128()` of the reference -> `build()` here (src/bn128.js:173-265)."""

    def __init__(self, lib=None, device=-1):
        self.lib = lib or _lib.load()
        self.device_info = self.lib.init(device)
        self._keys = {}

    # --- src/bn128.js:353-383 ---
    def g1_multiexp(self, scalars, points, shard=None):
        """shard=(rank, world): only the Pippenger windows w % world == rank (a partial sum, see g1_sum)."""
        return self._multiexp(1, scalars, points, shard)

    # --- src/bn128.js:385-415 ---
    def g2_multiexp(self, scalars, points, shard=None):
        return self._multiexp(2, scalars, points, shard)

    def _multiexp(self, g, scalars, points, shard):
        s, ns = _ro(scalars)
        p, np_ = _ro(points)
        n = ns // 32
        sz = 64 if g == 1 else 128
        if np_ < n * sz:
            raise ValueError("points buffer shorter than n*%d bytes" % sz)
        out = (C.c_uint8 * (96 if g == 1 else 192))()
        rank, world = shard or (0, 1)
        fn = self.lib.c.wsnark_g1_msm_windows if g == 1 else self.lib.c.wsnark_g2_msm_windows
        self.lib.check(fn(s, p, n, rank, world, out))
        return bytes(out)

    # --- the gather loop of src/bn128.js:374-382 / 406-414: EC sum of Jacobian partials ---
    def g1_sum(self, partials):
        b, n = _ro(partials)
        out = (C.c_uint8 * 96)()
        self.lib.check(self.lib.c.wsnark_g1_sum(b, n // 96, out))
        return bytes(out)

    def g2_sum(self, partials):
        b, n = _ro(partials)
        out = (C.c_uint8 * 192)()
        self.lib.check(self.lib.c.wsnark_g2_sum(b, n // 192, out))
        return bytes(out)

    # --- device-resident variants (pointers from torch tensors / hipMalloc) ---
    def g1_multiexp_dev(self, d_scalars, d_points, n, stream=None, shard=None):
        out = (C.c_uint8 * 96)()
        rank, world = shard or (0, 1)
        self.lib.check(self.lib.c.wsnark_g1_msm_windows_dev(d_scalars, d_points, n, rank, world, out, stream))
        return bytes(out)

    def g2_multiexp_dev(self, d_scalars, d_points, n, stream=None, shard=None):
        out = (C.c_uint8 * 192)()
        rank, world = shard or (0, 1)
        self.lib.check(self.lib.c.wsnark_g2_msm_windows_dev(d_scalars, d_points, n, rank, world, out, stream))
        return bytes(out)

    def fft_dev(self, d_buf, n, odd=0, inverse=False, stream=None):
        self.lib.check(self.lib.c.wsnark_fr_ntt_dev(d_buf, n, int(odd), 1 if inverse else 0, stream))

    def groth16GenProof_dev(self, d_witness, witness_len, key, r=None, s=None, stream=None):
        out = (C.c_uint8 * 384)()
        rb = _ro(r)[0] if r is not None else None
        sb = _ro(s)[0] if s is not None else None
        self.lib.check(self.lib.c.wsnark_groth16_prove_dev(key._h, d_witness, witness_len, rb, sb, out, stream))
        return proof_from_bytes(bytes(out))

    # --- src/bn128.js:569-578 (worker CALC_H :126-166) ---
    def calcH(self, signals, polsA, polsB, nSignals, domainSize):
        s, _ = _ro(signals)
        a, la = _ro(polsA)
        b, lb = _ro(polsB)
        out = (C.c_uint8 * (domainSize * 32))()
        self.lib.check(self.lib.c.wsnark_calc_h(s, a, la, b, lb, nSignals, domainSize, out))
        return bytes(out)

    # --- fft_fft / fft_ifft (src/build_fft.js:159-221) on Montgomery Fr elements ---
    def fft(self, data, odd=0, inverse=False):
        b, n = _buf(data)
        self.lib.check(self.lib.c.wsnark_fr_ntt(b, n // 32, int(odd), 1 if inverse else 0))
        return bytes(b)[:n]

    def ifft(self, data, odd=0):
        return self.fft(data, odd, True)

    def toMontgomeryN(self, data):
        b, n = _buf(data)
        self.lib.check(self.lib.c.wsnark_fr_to_montgomery(b, b, n // 32))
        return bytes(b)[:n]

    def fromMontgomeryN(self, data):
        b, n = _buf(data)
        self.lib.check(self.lib.c.wsnark_fr_from_montgomery(b, b, n // 32))
        return bytes(b)[:n]

    # --- synthetic-input helper (no reference counterpart): scalars[i] * generator, affine ---
    def mul_base(self, g, scalars, base=None):
        s, ns = _ro(scalars)
        n = ns // 32
        sz = 64 if g == 1 else 128
        b, _ = _ro(base if base is not None else (G1_GEN if g == 1 else G2_GEN))
        out = (C.c_uint8 * max(n * sz, 1))()
        fn = self.lib.c.wsnark_g1_mul_base_batch if g == 1 else self.lib.c.wsnark_g2_mul_base_batch
        self.lib.check(fn(b, s, n, out))
        return bytes(out)[: n * sz]

    def load_points(self, g, points):
        """Make a point set resident as fixed-base tables (no reference counterpart): see ResidentPoints."""
        return ResidentPoints(self.lib, g, points)

    def load_key(self, pkey=None, sections=None, shard=None, h_interleave_log=0, wait_tables=True, path=None):
        return ProvingKey(self.lib, pkey, sections, shard, h_interleave_log, wait_tables, path)

    def key_file_info(self, path):
        """Header of a key file (no GPU work): {n_vars, n_public, domain, file_bytes, format: 'proving_key.bin' | 'WSNARK64'}."""
        nv, npub, dom, nb, fmt = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_int()
        self.lib.check(self.lib.c.wsnark_pkey_file_info(os.fsencode(path), C.byref(nv), C.byref(npub), C.byref(dom), C.byref(nb), C.byref(fmt)))
        return {"n_vars": nv.value, "n_public": npub.value, "domain": dom.value, "file_bytes": nb.value,
                "format": {1: "proving_key.bin", 2: "WSNARK64"}.get(fmt.value, "?")}

    def h_multiexp_dev(self, key, d_h_slice, n, stream=None):
        """The H sum of one rank against the key handle's resident hExps share (wsnark_pkey_h_msm_dev)."""
        out = (C.c_uint8 * 96)()
        self.lib.check(self.lib.c.wsnark_pkey_h_msm_dev(key._h, d_h_slice, n, out, stream))
        return bytes(out)

    # --- multi-GPU proving: per-rank partial sums + host-side finish (include/wsnark.h) ---
    def groth16_prove_partial(self, signals, key, shard=(0, 1), skip_h=False):
        """shard=(rank, world): this rank's 576-byte record of partial sums (windows w % world == rank).
        skip_h: leave CALC_H and the H sum to the caller (WSNARK_PARTIAL_SKIP_H; the H slot is infinity)."""
        w, nw = _ro(signals)
        out = (C.c_uint8 * 576)()
        self.lib.check(self.lib.c.wsnark_groth16_prove_partial(key._h, w, nw, shard[0], shard[1], 1 if skip_h else 0, out))
        return bytes(out)

    def groth16_prove_partial_dev(self, d_witness, witness_len, key, shard=(0, 1), stream=None, skip_h=False):
        out = (C.c_uint8 * 576)()
        self.lib.check(self.lib.c.wsnark_groth16_prove_partial_dev(key._h, d_witness, witness_len, shard[0], shard[1],
                                                                    1 if skip_h else 0, out, stream))
        return bytes(out)

    def last_blinding(self):
        """(r, s) of the last proof assembled by this thread -- the reference's this._pr / this._ps (src/bn128.js:662-664)."""
        r, s = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
        self.lib.check(self.lib.c.wsnark_last_blinding(r, s))
        return bytes(r), bytes(s)

    def groth16_prove_finish(self, key, partials, r=None, s=None):
        p, n = _ro(partials)
        out = (C.c_uint8 * 384)()
        rb = _ro(r)[0] if r is not None else None
        sb = _ro(s)[0] if s is not None else None
        self.lib.check(self.lib.c.wsnark_groth16_prove_finish(key._h, p, n // 576, rb, sb, out))
        return proof_from_bytes(bytes(out))

    # --- src/bn128.js:580-720 ---
    def groth16GenProof(self, signals, pkey, r=None, s=None):
        """signals: witness.bin bytes; pkey: proving_key.bin bytes or a ProvingKey.
        r, s: optional 32-byte blinding values (the reference draws them with
        crypto.randomBytes, src/bn128.js:642-661). Returns {pi_a, pi_b, pi_c} of decimal strings."""
        key = pkey if isinstance(pkey, ProvingKey) else ProvingKey(self.lib, pkey)
        w, nw = _ro(signals)
        out = (C.c_uint8 * 384)()
        rb = _ro(r)[0] if r is not None else None
        sb = _ro(s)[0] if s is not None else None
        if (r is not None and len(r) != 32) or (s is not None and len(s) != 32):
            raise ValueError("r and s must be 32 bytes")
        self.lib.check(self.lib.c.wsnark_groth16_prove(key._h, w, nw, rb, sb, out))
        if key is not pkey:
            key.free()
        return proof_from_bytes(bytes(out))

    def groth16GenProof_hostptr(self, h_witness, witness_len, key, r=None, s=None):
        """The same call with the witness given as a raw HOST address (e.g. a pinned torch tensor's data_ptr(): a source the
        runtime already knows as pinned is DMA'd in place, chunk by chunk, without the staging copy)."""
        out = (C.c_uint8 * 384)()
        rb = _ro(r)[0] if r is not None else None
        sb = _ro(s)[0] if s is not None else None
        self.lib.check(self.lib.c.wsnark_groth16_prove(key._h, C.c_void_p(h_witness), witness_len, rb, sb, out))
        return proof_from_bytes(bytes(out))

    # --- src/bn128.js:722-791 ---
    def groth16Verify(self, verificationKey, input, proof):
        """verificationKey: the snarkjs "groth" verification_key.json object (vk_alfa_1, vk_beta_2, vk_gamma_2, vk_delta_2,
        IC; vk_alfabeta_12 is not needed); input: public signals (decimal strings / ints, a single value is wrapped like
        the reference does, :724-728); proof: {pi_a, pi_b, pi_c} of decimal strings.  Returns True / False."""
        return groth16_verify(self.lib, verificationKey, input, proof)

    def terminate(self):  # src/bn128.js:562-566
        self.lib.shutdown()


def proof_from_bytes(b):
    """bin2g1 / bin2g2 of the reference (src/bn128.js:319-351, 714-718)."""
    v = [str(int.from_bytes(b[i:i + 32], "little")) for i in range(0, 384, 32)]
    return {"pi_a": v[0:3], "pi_b": [v[3:5], v[5:7], v[7:9]], "pi_c": v[9:12]}


def proof_to_bytes(proof):
    """Inverse of proof_from_bytes: {pi_a, pi_b, pi_c} of decimal strings -> the 384 bytes wsnark_groth16_prove writes."""
    le = lambda v: int(v).to_bytes(32, "little")
    a, b, c = proof["pi_a"], proof["pi_b"], proof["pi_c"]
    return (b"".join(le(x) for x in a) + b"".join(le(x) for pair in b for x in pair) + b"".join(le(x) for x in c))


def vk_to_bytes(vk, n_inputs):
    le = lambda v: int(v).to_bytes(32, "little")
    g1 = lambda p: le(p[0]) + le(p[1])
    g2 = lambda p: le(p[0][0]) + le(p[0][1]) + le(p[1][0]) + le(p[1][1])
    if len(vk["IC"]) < n_inputs + 1:
        raise ValueError("verification key has %d IC points, %d inputs given" % (len(vk["IC"]), n_inputs))
    return (g1(vk["vk_alfa_1"]) + g2(vk["vk_beta_2"]) + g2(vk["vk_gamma_2"]) + g2(vk["vk_delta_2"])
            + b"".join(g1(p) for p in vk["IC"][:n_inputs + 1]))


def groth16_verify(lib, verificationKey, input, proof):
    """Bn128.groth16Verify (src/bn128.js:722-791) over wsnark_groth16_verify: host arithmetic, no GPU needed."""
    if input is None:
        input = []
    elif not isinstance(input, (list, tuple)):
        input = [input]
    vals = [int(x) for x in input]
    if any(v < 0 or v >= 1 << 256 for v in vals):
        return False
    vkb = vk_to_bytes(verificationKey, len(vals))
    inp = b"".join(v.to_bytes(32, "little") for v in vals)
    valid = C.c_int(0)
    lib.check(lib.c.wsnark_groth16_verify(vkb, len(vkb), inp if vals else None, len(vals), proof_to_bytes(proof), C.byref(valid)))
    return bool(valid.value)


def build(lib=None, device=-1):
    return Bn128(lib, device)


_singleton = None


def groth16GenProof(witness, provingKey, cb=None):
    """main_bn128.js:26-39 (window.groth16GenProof): optional node-style callback."""
    global _singleton
    try:
        if _singleton is None:
            _singleton = build()
        proof = _singleton.groth16GenProof(witness, provingKey)
    except Exception as e:  # noqa: BLE001 - mirrors cb(err)
        if cb:
            return cb(e, None)
        raise
    return cb(None, proof) if cb else proof


genZKSnarkProof = groth16GenProof  # README.md:28-30 name
