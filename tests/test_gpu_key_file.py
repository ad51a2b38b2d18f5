"""Key files on the MI355X (round 6; VERDICT r5 "next round" 3): a 2^22 key (2.5 GB) written as the WSNARK64 container and loaded BY
PATH -- whole, as a group of two contexts, as eight shards rank after rank, as one rank of the native distributed prover, and through
the Node drop-in -- proofs == the toxic-waste closed form == the sections loader's proof; a shard load's resident-set rise stays
under a quarter of the key (the loader maps the file and reads only the share's pages)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    import __graft_entry__
    __graft_entry__.ensure_built()
    import wasmsnark_amd
    from wasmsnark_amd import formats, synth
    bn = wasmsnark_amd.build(device=0)
    logd = int(os.environ.get("WSNARK_TEST_KEYFILE_LOG", "22"))
    circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=11)
    sec, _ = circ.build_sections()
    d = tmp_path_factory.mktemp("keyfile")
    path = str(d / "key.wsnark64")
    nbytes = formats.write_key_container(sec, path)
    wit = circ.witness_bin()
    wpath = str(d / "witness.bin")
    open(wpath, "wb").write(wit)
    r32, s32 = bytes(range(32)), bytes(range(64, 96))
    want = circ.expected_proof(r32, s32)
    yield {"bn": bn, "sec": sec, "path": path, "bytes": nbytes, "wit": wit, "wpath": wpath, "r": r32, "s": s32, "want": want, "logd": logd, "dir": str(d)}
    circ.free()


def test_whole_key_from_the_file_equals_sections_loader_and_closed_form(big):
    bn = big["bn"]
    info = bn.key_file_info(big["path"])
    assert info["format"] == "WSNARK64" and info["file_bytes"] == big["bytes"] and info["domain"] == 1 << big["logd"]
    k_file = bn.load_key(path=big["path"])
    k_sec = bn.load_key(sections=big["sec"])
    p_file = bn.groth16GenProof(big["wit"], k_file, r=big["r"], s=big["s"])
    p_sec = bn.groth16GenProof(big["wit"], k_sec, r=big["r"], s=big["s"])
    assert p_file == p_sec == big["want"]
    assert k_file.table == k_sec.table
    k_file.free(); k_sec.free()


def test_group_of_two_and_eight_shards_from_the_file(big):
    bn = big["bn"]
    from wasmsnark_amd import bn128
    g = bn128.Group(lib=bn.lib, devices=[0, 0])
    try:
        gk = g.load_key(path=big["path"])
        assert g.groth16GenProof(big["wit"], gk, r=big["r"], s=big["s"]) == big["want"]
        gk.free()
    finally:
        g.terminate()
    recs, sh = b"", None
    for rank in range(8):
        sh = bn.load_key(path=big["path"], shard=(rank, 8))
        recs += bn.groth16_prove_partial(big["wit"], sh, shard=(rank, 8))
        if rank < 7:
            sh.free()
    assert bn.groth16_prove_finish(sh, recs, r=big["r"], s=big["s"]) == big["want"]
    sh.free()


def test_native_dist_prover_world_of_one_from_the_file(big):
    import torch
    from wasmsnark_amd import dist as wdist
    bn = big["bn"]
    npv = wdist.NativeDistProver(bn, path=big["path"], device=torch.device("cuda:0"))
    d_w = torch.frombuffer(bytearray(big["wit"]), dtype=torch.uint8).cuda()
    assert npv.prove(d_w.data_ptr(), len(big["wit"]), r=big["r"], s=big["s"]) == big["want"]
    npv.key.free()


def test_shard_load_keeps_the_resident_set_small(big):
    """One of eight shards (in the interleaved hExps layout the distributed prover uses): the process's resident set may rise by at most
    a quarter of the key while it loads -- the sections loader needed the WHOLE key in host memory on every rank."""
    from wasmsnark_amd import formats
    small = os.path.join(big["dir"], "small.bin")
    open(small, "wb").write(open(os.path.join(ROOT, "tests", "golden", "keys", "t6.pkey.bin"), "rb").read())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "key_file_rss_worker.py"), big["path"], "3", "8", str(big["logd"] // 2), small],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print("shard load RSS:", r)
    assert r["shard"]["world"] == 8 and r["shard"]["rank"] == 3
    assert r["rise"] <= big["bytes"] // 4, r
    del formats


@pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"), reason="node / N-API headers not available")
def test_node_drop_in_proves_from_the_file(big):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "wasmsnark_amd", "js"), "-s"])
    wj = os.path.join(big["dir"], "want.json")
    json.dump(big["want"], open(wj, "w"))
    for devs in (None, "0,0"):
        cmd = ["node", os.path.join(ROOT, "tests", "node_key_file_check.js"), big["path"], big["wpath"], wj, big["r"].hex(), big["s"].hex()] + ([devs] if devs else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout + out.stderr
        r = json.loads(out.stdout.strip().splitlines()[-1])
        print("node key file:", r)
        assert r["ok"] and r["info"]["format"] == "WSNARK64"
