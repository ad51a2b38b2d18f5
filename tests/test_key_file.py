"""Key FILES (round 6): the reference's proving_key.bin by path, and WSNARK64 -- the same sections behind
64-bit offsets, the container for keys beyond the 4 GiB of the reference's u32 offsets (/root/reference tools/buildpkey.js:133-139).
CPU: the loader on the thread-emulator build (mapping, bounds checks, shards read from the file, the group loader, release of the
pages); the writers of formats.py and js/formats.js produce the same bytes."""
import json
import os
import shutil
import struct
import subprocess

import pytest

from conftest import GOLDEN, ROOT, load_golden
from emul_util import emul_bn128
from wasmsnark_amd import bn128, formats


def _key(name):
    rd = lambda ext: open(os.path.join(GOLDEN, "keys", name + ext), "rb").read()
    return rd(".pkey.bin"), rd(".witness.bin")


def test_container_round_trip_and_layout(tmp_path):
    pk, _ = _key("t6")
    sec = formats.pkey_bin_to_sections(pk)
    p = str(tmp_path / "k.wsnark64")
    n = formats.write_key_container(sec, p)
    raw = open(p, "rb").read()
    assert len(raw) == n and raw[:8] == b"WSNARK64"
    ver, hb, nv, npub, dom, z = struct.unpack_from("<6I", raw, 8)
    assert (ver, hb, nv, npub, dom, z) == (1, 608, sec["n_vars"], sec["n_public"], sec["domain"], 0)
    offs = struct.unpack_from("<10Q", raw, 32)
    assert offs[9] == n and all(o % 4096 == 0 for o in (offs[0], offs[2], offs[4], offs[5], offs[6], offs[7], offs[8]))
    # same section order as tools/buildpkey.js:166-186, same bytes as the reference-format file
    assert offs[0] < offs[2] < offs[4] < offs[5] < offs[6] < offs[7] < offs[8]
    back = formats.read_key_container(p)
    for k, v in sec.items():
        assert back[k] == v, k
    assert raw[160:608] == pk[40:488]


def test_load_file_both_formats_and_shards(tmp_path):
    bn = emul_bn128()
    for name in ("t6", "t3"):
        pk, wit = _key(name)
        p_bin, p_64 = str(tmp_path / (name + ".bin")), str(tmp_path / (name + ".wsnark64"))
        open(p_bin, "wb").write(pk)
        formats.pkey_bin_to_container(pk, p_64)
        assert bn.key_file_info(p_bin)["format"] == "proving_key.bin" and bn.key_file_info(p_64)["format"] == "WSNARK64"
        assert bn.key_file_info(p_64)["file_bytes"] == os.path.getsize(p_64)
        cases = load_golden("proofs.json")[name]
        for path in (p_bin, p_64):
            k = bn.load_key(path=path)
            for c in cases:
                assert bn.groth16GenProof(wit, k, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]     # the REFERENCE's proofs
            k.free()
        c = cases[1]
        for world in (2, 3, 8):            # uneven and (t3: 8 > nVars / 2) nearly empty shares, each read from the file
            recs, sh = b"", None
            for rank in range(world):
                sh = bn.load_key(path=p_64, shard=(rank, world))
                assert (sh.shard["rank"], sh.shard["world"]) == (rank, world)
                recs += bn.groth16_prove_partial(wit, sh, shard=(rank, world))
            assert bn.groth16_prove_finish(sh, recs, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]


def test_group_loads_its_shards_from_one_file(tmp_path):
    bn = emul_bn128()
    pk, wit = _key("t6")
    p_64 = str(tmp_path / "k.wsnark64")
    formats.pkey_bin_to_container(pk, p_64)
    c = load_golden("proofs.json")["t6"][0]
    for world in (2, 3, 4):
        g = bn128.Group(lib=bn.lib, devices=[0] * world)
        try:
            key = g.load_key(path=p_64)
            assert key.world == world
            assert g.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]
        finally:
            g.terminate()


def test_bad_files_are_format_errors_not_crashes(tmp_path):
    bn = emul_bn128()
    pk, _ = _key("t6")
    good = str(tmp_path / "good")
    n = formats.pkey_bin_to_container(pk, good)
    raw = bytearray(open(good, "rb").read())

    def load(b):
        p = str(tmp_path / "bad")
        open(p, "wb").write(b)
        with pytest.raises(Exception) as e:
            bn.load_key(path=p)
        return str(e.value)
    assert "FORMAT" in load(raw[:600])                                   # shorter than the header
    assert "FORMAT" in load(raw[:n - 64])                                # truncated: length word disagrees
    v2 = bytearray(raw); struct.pack_into("<I", v2, 8, 2)
    assert "FORMAT" in load(v2)                                          # unknown version
    for word in range(9):                                                # every offset / length pushed out of the file
        b = bytearray(raw); struct.pack_into("<Q", b, 32 + 8 * word, n + 4096)
        assert "FORMAT" in load(b), word
    b = bytearray(raw); struct.pack_into("<I", b, 16, 1 << 30)           # nVars that the sections cannot hold
    assert "FORMAT" in load(b)
    with pytest.raises(Exception):
        bn.load_key(path=str(tmp_path / "does-not-exist"))
    with pytest.raises(Exception):
        bn.load_key(path=str(tmp_path))                                  # a directory
    # ... and the library is still healthy
    k = bn.load_key(path=good); k.free()


def test_streamed_matrices_equal_the_one_piece_upload(tmp_path):
    """A mapped file's record streams go up in pieces AS the header walk passes them, every piece's pages handed back at once
    (calch.hip: pols_to_csr with the loader's release hook); the in-memory loader walks, then uploads in one piece.  Same matrices,
    same proofs -- checked with pieces of 1 KiB, so that the small keys' streams span many of them."""
    bn = emul_bn128()
    bn.lib.tune("POLS_PIECE_KB", 1)
    try:
        for name in ("t6", "t3"):
            pk, wit = _key(name)
            p_64 = str(tmp_path / (name + ".wsnark64"))
            formats.pkey_bin_to_container(pk, p_64)
            assert len(formats.pkey_bin_to_sections(pk)["polsA"]) > (3 << 10 if name == "t6" else 0)
            k = bn.load_key(path=p_64)
            for c in load_golden("proofs.json")[name]:
                assert bn.groth16GenProof(wit, k, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"]
            k.free()
    finally:
        bn.lib.tune("POLS_PIECE_KB", None)


@pytest.mark.skipif(shutil.which("node") is None, reason="node not available")
def test_js_writer_writes_the_same_bytes(tmp_path):
    pk, _ = _key("t6")
    p_py, p_js, p_in = str(tmp_path / "py"), str(tmp_path / "js"), str(tmp_path / "in.bin")
    open(p_in, "wb").write(pk)
    formats.pkey_bin_to_container(pk, p_py)
    code = "const f=require('%s/wasmsnark_amd/js/formats.js');f.pkeyBinToContainer(require('fs').readFileSync('%s'),'%s')" % (ROOT, p_in, p_js)
    subprocess.check_call(["node", "-e", code])
    assert open(p_py, "rb").read() == open(p_js, "rb").read()
