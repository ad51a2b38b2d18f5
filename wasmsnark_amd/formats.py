"""snarkjs JSON  <->  wasmsnark binary formats (proving_key.bin, witness.bin) and proof JSON.

Host-side data formats either side of the hot path.  Behaviour restated from
the reference's converters:

  * ``tools/buildpkey.js:124-186``  (proving_key.json -> proving_key.bin; layout notes at :192-240)
  * ``tools/buildwitness.js:36-69`` (witness.json -> witness.bin)
  * ``src/bn128.js:319-351,714-718`` (proof -> decimal strings)

Layout of proving_key.bin (all little-endian):

  u32 nVars, nPublic, domainSize, pPolsA, pPolsB, pPointsA, pPointsB1, pPointsB2, pPointsC, pHExps
  alfa1, beta1, delta1 (G1: x, y; 2 x 32 B, Montgomery mod q)   beta2, delta2 (G2: x.c0, x.c1, y.c0, y.c1)
  polsA[nVars], polsB[nVars] : u32 ncoefs, then ncoefs x (u32 constraint index, 32 B coefficient in
                               Montgomery form mod r), indices in ascending order (JS key order)
  A[nVars], B1[nVars] (G1), B2[nVars] (G2), C[nPublic+1 .. nVars-1] (G1), hExps[domainSize] (G1)

A point at infinity is ``["0", "1", "0"]`` in snarkjs JSON; the converter writes its first two coordinates
like any other point, so it becomes x == 0, which the prover treats as infinity (SURVEY.md fact list).

Command line (the reference's two tools under their own names):

    python -m wasmsnark_amd.formats buildpkey    -i proving_key.json -o proving_key.bin
    python -m wasmsnark_amd.formats buildwitness -i witness.json     -o witness.bin
    python -m wasmsnark_amd.formats dumppkey     -i proving_key.bin  -o proving_key.json
    python -m wasmsnark_amd.formats pkey64       -i proving_key.bin  -o proving_key.wsnark64     (the u64-offset container, below)
"""
from __future__ import annotations

import json
import struct
import sys

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_MONT = 1 << 256
_MONT_INV_Q = pow(_MONT, -1, Q)
_MONT_INV_R = pow(_MONT, -1, R)


class FormatError(ValueError):
    pass


def _int(v) -> int:
    if isinstance(v, bool):
        raise FormatError("boolean where an integer was expected")
    if isinstance(v, int):
        return v
    if isinstance(v, str) and v.isdigit():
        return int(v)
    raise FormatError("not a decimal integer: %r" % (v,))


def _le32(v: int) -> bytes:
    # tools/buildpkey.js:54-59 writes the low 256 bits, eight u32 words, little-endian
    return (v & ((1 << 256) - 1)).to_bytes(32, "little")


def _mont_q(v) -> bytes:
    return _le32(_int(v) * _MONT % Q)


def _mont_r(v) -> bytes:
    return _le32(_int(v) * _MONT % R)


def _g1(p) -> bytes:
    return _mont_q(p[0]) + _mont_q(p[1])


def _g2(p) -> bytes:
    return _mont_q(p[0][0]) + _mont_q(p[0][1]) + _mont_q(p[1][0]) + _mont_q(p[1][1])


def _js_key_order(keys):
    """Object.keys order of a JS object whose keys came from JSON: canonical array indices (``"0"``,
    ``"17"``; no leading zeros, < 2^32 - 1) ascending first, then the other keys in insertion order."""
    idx, other = [], []
    for k in keys:
        if k.isdigit() and (k == "0" or k[0] != "0") and int(k) < 4294967295:
            idx.append(k)
        else:
            other.append(k)
    idx.sort(key=int)
    return idx + other


def _pol(p) -> bytes:
    if not isinstance(p, dict):
        raise FormatError("polynomial must be an object {constraint index: coefficient}")
    keys = _js_key_order(list(p.keys()))
    out = [struct.pack("<I", len(keys))]
    for k in keys:
        if not k.isdigit():
            raise FormatError("polynomial key is not a constraint index: %r" % k)
        out.append(struct.pack("<I", int(k) & 0xFFFFFFFF))
        out.append(_mont_r(p[k]))
    return b"".join(out)


def pkey_json_to_bin(pk: dict) -> bytes:
    """proving_key.json (snarkjs "groth" key, parsed) -> proving_key.bin bytes (tools/buildpkey.js:124-186)."""
    n_vars, n_public, domain = _int(pk["nVars"]), _int(pk["nPublic"]), _int(pk["domainSize"])
    for name in ("polsA", "polsB", "A", "B1", "B2", "C"):
        if len(pk[name]) < n_vars:
            raise FormatError("%s has %d entries, nVars is %d" % (name, len(pk[name]), n_vars))
    if len(pk["hExps"]) < domain:
        raise FormatError("hExps has %d entries, domainSize is %d" % (len(pk["hExps"]), domain))
    head = b"".join([_g1(pk["vk_alfa_1"]), _g1(pk["vk_beta_1"]), _g1(pk["vk_delta_1"]),
                     _g2(pk["vk_beta_2"]), _g2(pk["vk_delta_2"])])
    pols_a = b"".join(_pol(pk["polsA"][i]) for i in range(n_vars))
    pols_b = b"".join(_pol(pk["polsB"][i]) for i in range(n_vars))
    pts_a = b"".join(_g1(pk["A"][i]) for i in range(n_vars))
    pts_b1 = b"".join(_g1(pk["B1"][i]) for i in range(n_vars))
    pts_b2 = b"".join(_g2(pk["B2"][i]) for i in range(n_vars))
    pts_c = b"".join(_g1(pk["C"][i]) for i in range(n_public + 1, n_vars))
    pts_h = b"".join(_g1(pk["hExps"][i]) for i in range(domain))
    offs, o = [], 40 + len(head)
    for sec in (pols_a, pols_b, pts_a, pts_b1, pts_b2, pts_c, pts_h):
        offs.append(o)
        o += len(sec)
    if o > 0xFFFFFFFF:
        raise FormatError("key is %d bytes: proving_key.bin addresses sections with u32 offsets "
                          "(use wsnark_pkey_load_sections / Bn128.load_key(sections=...) instead)" % o)
    return b"".join([struct.pack("<10I", n_vars, n_public, domain, *offs), head,
                     pols_a, pols_b, pts_a, pts_b1, pts_b2, pts_c, pts_h])


def witness_json_to_bin(w) -> bytes:
    """witness.json (list of decimal strings) -> witness.bin: nVars x 32 B plain little-endian
    (tools/buildwitness.js:36-69; values are written as they are, not reduced)."""
    if not isinstance(w, list):
        raise FormatError("witness must be a JSON array")
    return b"".join(_le32(_int(v)) for v in w)


def witness_bin_to_json(b: bytes) -> list:
    if len(b) % 32:
        raise FormatError("witness.bin length is not a multiple of 32")
    return [str(int.from_bytes(b[i:i + 32], "little")) for i in range(0, len(b), 32)]


# ---- the inverse of buildpkey (not in the reference; handy for inspection, tests and interop) ----

def _unmont_q(b: bytes) -> str:
    return str(int.from_bytes(b, "little") * _MONT_INV_Q % Q)


def _unmont_r(b: bytes) -> str:
    return str(int.from_bytes(b, "little") * _MONT_INV_R % R)


def _g1_json(b: bytes):
    # x == 0 is the format's infinity; y is kept as stored so that dump -> build reproduces the bytes
    x, y = _unmont_q(b[:32]), _unmont_q(b[32:64])
    return [x, y, "0" if x == "0" else "1"]


def _g2_json(b: bytes):
    c = [_unmont_q(b[i:i + 32]) for i in range(0, 128, 32)]
    inf = c[0] == "0" and c[1] == "0"
    return [[c[0], c[1]], [c[2], c[3]], ["0", "0"] if inf else ["1", "0"]]


def pkey_bin_to_json(b: bytes) -> dict:
    """proving_key.bin -> the snarkjs JSON fields the prover uses (C[0..nPublic] are written as infinity:
    the binary format does not carry them)."""
    if len(b) < 40 + 448:
        raise FormatError("proving key shorter than its header")
    n_vars, n_public, domain, p_a, p_b, p_pa, p_pb1, p_pb2, p_pc, p_h = struct.unpack_from("<10I", b, 0)

    def pols(off, end):
        out = []
        for _ in range(n_vars):
            if off + 4 > end:
                raise FormatError("polynomial section overruns")
            (nc,) = struct.unpack_from("<I", b, off)
            off += 4
            if off + 36 * nc > end:
                raise FormatError("polynomial record overruns its section")
            d = {}
            for _j in range(nc):
                (idx,) = struct.unpack_from("<I", b, off)
                d[str(idx)] = _unmont_r(b[off + 4:off + 36])
                off += 36
            out.append(d)
        return out

    def g1s(off, n):
        if off + 64 * n > len(b):
            raise FormatError("point section overruns the key")
        return [_g1_json(b[off + 64 * i:off + 64 * i + 64]) for i in range(n)]

    def g2s(off, n):
        if off + 128 * n > len(b):
            raise FormatError("point section overruns the key")
        return [_g2_json(b[off + 128 * i:off + 128 * i + 128]) for i in range(n)]

    n_c = n_vars - n_public - 1
    return {
        "protocol": "groth", "nVars": n_vars, "nPublic": n_public, "domainSize": domain,
        "domainBits": max(domain.bit_length() - 1, 0),
        "vk_alfa_1": _g1_json(b[40:104]), "vk_beta_1": _g1_json(b[104:168]), "vk_delta_1": _g1_json(b[168:232]),
        "vk_beta_2": _g2_json(b[232:360]), "vk_delta_2": _g2_json(b[360:488]),
        "polsA": pols(p_a, p_b), "polsB": pols(p_b, p_pa),
        "A": g1s(p_pa, n_vars), "B1": g1s(p_pb1, n_vars), "B2": g2s(p_pb2, n_vars),
        "C": [["0", "1", "0"]] * (n_public + 1) + g1s(p_pc, n_c),
        "hExps": g1s(p_h, domain),
    }


def pkey_bin_to_sections(b: bytes) -> dict:
    """proving_key.bin -> the separate sections wsnark_pkey_load_sections / wsnark_pkey_load_shard take (true section
    bounds from the header, tools/buildpkey.js:124-186; the reference itself slices over-long, src/bn128.js:592-593)."""
    if len(b) < 488:
        raise FormatError("proving key shorter than its fixed header")
    nv, npub, dom, pPA, pPB, pA, pB1, pB2, pC, pH = struct.unpack_from("<10I", b, 0)
    if npub + 1 > nv or not (488 <= pPA <= pPB <= pA) or pH + dom * 64 > len(b) or pB2 + nv * 128 > len(b):
        raise FormatError("proving key: section offsets out of range")
    return {"n_vars": nv, "n_public": npub, "domain": dom, "alfa1": b[40:104], "beta1": b[104:168], "delta1": b[168:232],
            "beta2": b[232:360], "delta2": b[360:488], "polsA": b[pPA:pPB], "polsB": b[pPB:pA],
            "pointsA": b[pA:pA + nv * 64], "pointsB1": b[pB1:pB1 + nv * 64], "pointsB2": b[pB2:pB2 + nv * 128],
            "pointsC": b[pC:pC + (nv - npub - 1) * 64], "pointsH": b[pH:pH + dom * 64]}


# ---- the u64-offset container for keys beyond proving_key.bin's 4 GiB (layout: csrc/keyfile.hip) ----
KEY_CONTAINER_MAGIC = b"WSNARK64"
_CONTAINER_HEADER = 608
_CONTAINER_ALIGN = 4096
_SECTION_ORDER = ("polsA", "polsB", "pointsA", "pointsB1", "pointsB2", "pointsC", "pointsH")      # tools/buildpkey.js:166-186


def write_key_container(sections: dict, path) -> int:
    """The sections of a key (the dict pkey_bin_to_sections returns / wsnark_pkey_load_sections takes; values may be bytes,
    memoryviews or numpy arrays of any size) -> a WSNARK64 file at `path`: the layout of proving_key.bin
    (tools/buildpkey.js:124-186, same section order, same bytes) behind 64-bit offsets.  Streams section by section; returns the
    file's length.  wsnark_pkey_load_file / Bn128.load_key(path=...) / loadKey(path) read it."""
    nv, npub, dom = int(sections["n_vars"]), int(sections["n_public"]), int(sections["domain"])
    views = {k: memoryview(sections[k]).cast("B") for k in _SECTION_ORDER}
    want = {"pointsA": nv * 64, "pointsB1": nv * 64, "pointsB2": nv * 128, "pointsC": (nv - npub - 1) * 64, "pointsH": dom * 64}
    for k, n in want.items():
        if len(views[k]) < n:
            raise FormatError("key container: section %s is shorter than its header-implied %d bytes" % (k, n))
        views[k] = views[k][:n]
    off, offs = _CONTAINER_HEADER, {}
    for k in _SECTION_ORDER:
        off = (off + _CONTAINER_ALIGN - 1) // _CONTAINER_ALIGN * _CONTAINER_ALIGN
        offs[k] = off
        off += len(views[k])
    total = off
    hdr = bytearray(_CONTAINER_HEADER)
    hdr[0:8] = KEY_CONTAINER_MAGIC
    struct.pack_into("<6I", hdr, 8, 1, _CONTAINER_HEADER, nv, npub, dom, 0)
    struct.pack_into("<10Q", hdr, 32, offs["polsA"], len(views["polsA"]), offs["polsB"], len(views["polsB"]), offs["pointsA"], offs["pointsB1"],
                     offs["pointsB2"], offs["pointsC"], offs["pointsH"], total)
    fixed = b"".join(bytes(memoryview(sections[k]).cast("B")) for k in ("alfa1", "beta1", "delta1", "beta2", "delta2"))
    if len(fixed) != 448:
        raise FormatError("key container: alfa1 / beta1 / delta1 are 64 bytes, beta2 / delta2 128 bytes")
    hdr[160:608] = fixed
    with open(path, "wb") as f:
        f.write(hdr)
        for k in _SECTION_ORDER:
            f.seek(offs[k])
            v = views[k]
            for lo in range(0, len(v), 64 << 20):       # (pieces: a 2 GiB section in one write() is cut short by the kernel)
                f.write(v[lo:lo + (64 << 20)])
        f.truncate(total)
    return total


def pkey_bin_to_container(b, path) -> int:
    """proving_key.bin bytes -> the same key as a WSNARK64 file (what one would do before sharding a key over ranks by file)."""
    return write_key_container(pkey_bin_to_sections(b), path)


def read_key_container(path) -> dict:
    """A WSNARK64 file -> its sections as bytes (tests, tools; the loader maps the file instead: wsnark_pkey_load_file)."""
    with open(path, "rb") as f:
        hdr = f.read(_CONTAINER_HEADER)
        if len(hdr) < _CONTAINER_HEADER or hdr[:8] != KEY_CONTAINER_MAGIC:
            raise FormatError("not a WSNARK64 key container")
        ver, hb, nv, npub, dom, _ = struct.unpack_from("<6I", hdr, 8)
        pPA, lPA, pPB, lPB, pA, pB1, pB2, pC, pH, total = struct.unpack_from("<10Q", hdr, 32)
        if ver != 1 or hb < _CONTAINER_HEADER:
            raise FormatError("key container: unknown version")
        f.seek(0, 2)
        if f.tell() != total:
            raise FormatError("key container: the file is not as long as its header says")

        def rd(off, n):
            f.seek(off)
            b = f.read(n)
            if len(b) != n:
                raise FormatError("key container: section out of range")
            return b
        return {"n_vars": nv, "n_public": npub, "domain": dom, "alfa1": hdr[160:224], "beta1": hdr[224:288], "delta1": hdr[288:352],
                "beta2": hdr[352:480], "delta2": hdr[480:608], "polsA": rd(pPA, lPA), "polsB": rd(pPB, lPB), "pointsA": rd(pA, nv * 64),
                "pointsB1": rd(pB1, nv * 64), "pointsB2": rd(pB2, nv * 128), "pointsC": rd(pC, (nv - npub - 1) * 64), "pointsH": rd(pH, dom * 64)}


def proof_from_bytes(p: bytes) -> dict:
    """The 384-byte proof record of wsnark_groth16_prove (12 plain LE 256-bit integers) -> the reference's
    proof object of decimal strings (src/bn128.js:714-718)."""
    if len(p) != 384:
        raise FormatError("proof record must be 384 bytes")
    v = [str(int.from_bytes(p[i:i + 32], "little")) for i in range(0, 384, 32)]
    return {"pi_a": v[0:3], "pi_b": [v[3:5], v[5:7], v[7:9]], "pi_c": v[9:12]}


def _main(argv) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m wasmsnark_amd.formats")
    ap.add_argument("tool", choices=["buildpkey", "buildwitness", "dumppkey", "dumpwitness", "pkey64"])
    ap.add_argument("-i", "--input")
    ap.add_argument("-o", "--output")
    a = ap.parse_args(argv)
    defaults = {"buildpkey": ("proving_key.json", "proving_key.bin"), "buildwitness": ("witness.json", "witness.bin"),
                "dumppkey": ("proving_key.bin", "proving_key.json"), "dumpwitness": ("witness.bin", "witness.json"),
                "pkey64": ("proving_key.bin", "proving_key.wsnark64")}
    src, dst = a.input or defaults[a.tool][0], a.output or defaults[a.tool][1]
    if a.tool == "pkey64":
        with open(src, "rb") as f:
            pkey_bin_to_container(f.read(), dst)
        return 0
    if a.tool in ("buildpkey", "buildwitness"):
        with open(src, "r") as f:
            obj = json.load(f)
        data = pkey_json_to_bin(obj) if a.tool == "buildpkey" else witness_json_to_bin(obj)
        with open(dst, "wb") as f:
            f.write(data)
    else:
        with open(src, "rb") as f:
            raw = f.read()
        obj = pkey_bin_to_json(raw) if a.tool == "dumppkey" else witness_bin_to_json(raw)
        with open(dst, "w") as f:
            json.dump(obj, f)
    return 0


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
