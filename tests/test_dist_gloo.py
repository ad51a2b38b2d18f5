"""N>1 path on CPU: two processes (gloo), each computes the partial MSM of its contiguous shard
(kernel sources under the CPU emulator), one all_gather of the 96/192-byte partials, local EC sum;
every rank must hold the full MSM (src/bn128.js:353-383 with workers = ranks)."""
import os
import random
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys, random
sys.path.insert(0, os.environ["WS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["WS_ROOT"], "tests"))
import torch.distributed as dist
from emul_util import emul_bn128
from oracle import pyoracle as orc
from wasmsnark_amd import dist as wd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
bn = emul_bn128()
assert bn.device_info.endswith("device=%d" % (rank % 2)), bn.device_info     # (two emulated devices: rank 1 works on device 1)
rnd = random.Random(42)
for g, n in ((1, 45), (2, 21)):
    sz = 64 if g == 1 else 128
    ks = b"".join(rnd.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n))
    pts = bn.mul_base(g, ks)
    sc = b"".join(rnd.randrange(1 << 256).to_bytes(32, "little") for _ in range(n))
    lo, hi = wd.shard_bounds(n, world, rank)
    f = bn.g1_multiexp if g == 1 else bn.g2_multiexp
    part = f(sc[lo * 32:hi * 32], pts[lo * sz:hi * sz])
    full = wd.sharded_msm(bn, g, part)
    want = orc.g_affine(g, orc.multiexp(g, "multiexp", sc, pts, n))
    assert full == want, (g, rank)
# north-star variant: every rank sees all pairs, owns the windows w % world == rank
for g, n in ((1, 45), (2, 21)):
    rnd2 = random.Random(7 + g)
    ks = b"".join(rnd2.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n))
    pts = bn.mul_base(g, ks)
    sc = b"".join(rnd2.randrange(1 << 256).to_bytes(32, "little") for _ in range(n))
    part = (bn.g1_multiexp if g == 1 else bn.g2_multiexp)(sc, pts, shard=(rank, world))
    full = wd.sharded_msm(bn, g, part)
    want = orc.g_affine(g, orc.multiexp(g, "multiexp", sc, pts, n))
    assert full == want, ("windows", g, rank)
    assert (bn.g1_multiexp if g == 1 else bn.g2_multiexp)(sc, pts) == want     # no mode left behind: the next call is whole
# whole proof, window-sharded: one 576-byte record per rank, one all_gather
import json
gold = os.path.join(os.environ["WS_ROOT"], "tests", "golden")
key = bn.load_key(open(os.path.join(gold, "keys", "t6.pkey.bin"), "rb").read())
wit = open(os.path.join(gold, "keys", "t6.witness.bin"), "rb").read()
for c in json.load(open(os.path.join(gold, "proofs.json")))["t6"]:
    got = wd.sharded_prove(bn, key, wit, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"]))
    assert got == c["proof"], ("sharded prove", rank)
# blinding left to the library: rank 0 draws, every rank must still assemble the SAME (valid) proof
got = wd.sharded_prove(bn, key, wit)
import torch
mine = torch.frombuffer(bytearray(json.dumps(got, sort_keys=True).encode().ljust(4096)), dtype=torch.uint8)
both = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(both[0], b) for b in both), "ranks disagree on the proof"
r_used, s_used = bn.last_blinding()
assert got == bn.groth16GenProof(wit, key, r=r_used, s=s_used)                 # == the single-process proof for those r, s
for v in (r_used, s_used):          # the reference's check of its draw (test/bn128_prover.js:65-71); rank 0 drew for everybody
    assert 96 <= bin(int.from_bytes(v, "little"))[2:].count("0") <= 160, "hamming weight of drawn blinding"
dist.barrier()
open(os.path.join(os.environ["WS_OUT"], "rank%d.ok" % rank), "w").write("ok")
'''


def test_sharded_msm_world2(tmp_path):
    from emul_util import emul_bn128
    emul_bn128()   # build the emulator library once, before the ranks race for it
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, WS_ROOT=ROOT, WS_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", WSNARK_EMUL_DEVICES="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_shard_bounds():
    from wasmsnark_amd.dist import shard_bounds
    # floor(n/W) each, remainder to the last; n < W gives the first W-1 workers nothing
    assert [shard_bounds(10, 4, r) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]
    assert [shard_bounds(3, 8, r) for r in range(8)][-1] == (0, 3)


def test_window_shards_sum_to_full_msm(orc):
    """single process: the partial sums of the window shards of any world size add up to the MSM"""
    import random
    from emul_util import emul_bn128
    bn = emul_bn128()
    rnd = random.Random(99)
    n = 70
    pts = bn.mul_base(1, b"".join(rnd.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n)))
    sc = b"".join(rnd.randrange(1 << 256).to_bytes(32, "little") for _ in range(n))
    want = orc.g_affine(1, orc.multiexp(1, "multiexp2", sc, pts, n))
    for world in (2, 3, 8, 70):
        parts = b"".join(bn.g1_multiexp(sc, pts, shard=(rank, world)) for rank in range(world))
        assert bn.g1_sum(parts) == want, world
    assert bn.g1_multiexp(sc, pts) == want
    import pytest
    for bad in ((1, 1), (5, 0), (3, 2)):
        with pytest.raises(Exception):
            bn.g1_multiexp(sc, pts, shard=bad)
