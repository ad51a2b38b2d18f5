"""Distributed four-step NTT (wasmsnark_amd/dist.py: dist_ntt) on CPU: two processes on gloo, kernel sources under the
thread emulator, against the pinned oracle's fft / ifft (reference semantics, src/build_fft.js:159-221): bit-exact for
2^4 .. 2^14, odd 0/1, forward and inverse; a chain (inverse, coset forward) without any re-layout in between; and whole
proofs through DistProver (distributed CALC_H + points-sharded H sum + window-sharded sums) against the reference's
golden proofs."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys, random
sys.path.insert(0, os.environ["WS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["WS_ROOT"], "tests"))
import torch, torch.distributed as dist
from emul_util import emul_bn128
from oracle import pyoracle as orc
from wasmsnark_amd import dist as wd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
bn = emul_bn128()
assert bn.device_info.endswith("device=%d" % (rank % 2)), bn.device_info     # (two emulated devices: rank 1 works on device 1)
def gather(y):
    parts = [torch.empty_like(y) for _ in range(world)]
    dist.all_gather(parts, y)
    return torch.cat(parts)
for bits in (4, 5, 10, 11, 14):
    n = 1 << bits
    rnd = random.Random(bits)
    x = orc.to_mont_n(b"".join(rnd.randrange(orc.R).to_bytes(32, "little") for _ in range(n)))
    full = torch.frombuffer(bytearray(x), dtype=torch.uint8)
    l1, l2 = wd.ntt_layout_split(bits, world)
    for odd in (0, 1):
        for inverse in (False, True):
            loc = wd.to_interleaved(full, l1, rank, world).clone()
            y = wd.dist_ntt(bn, loc, bits, odd=odd, inverse=inverse)
            got = wd.from_interleaved(gather(y), l2).numpy().tobytes()
            assert got == orc.fft(x, n, odd, inverse=inverse), (bits, odd, inverse, rank)
# three vectors through ONE transform / one exchange (dist_ntt(stack=3)) == three separate transforms
bits = 8
n = 1 << bits
l1, l2 = wd.ntt_layout_split(bits, world)
vs = [orc.to_mont_n(b"".join(random.Random(70 + j).randrange(orc.R).to_bytes(32, "little") for _ in range(n))) for j in range(3)]
stack = torch.cat([wd.to_interleaved(torch.frombuffer(bytearray(v), dtype=torch.uint8), l1, rank, world) for v in vs]).clone()
for odd, inverse in ((1, False), (0, True)):
    y = wd.dist_ntt(bn, stack.clone(), bits, odd=odd, inverse=inverse, stack=3)
    per = y.numel() // 3
    for j in range(3):
        got = wd.from_interleaved(gather(y[j * per:(j + 1) * per].contiguous()), l2).numpy().tobytes()
        assert got == orc.fft(vs[j], n, odd, inverse=inverse), ("stack", j, odd, inverse, rank)
# a chain as in CALC_H (src/bn128.js:150-153): coefficients = ifft(x), then evaluations on the odd coset = fft(., odd=1);
# with n1 == n2 the output layout of one transform is the input layout of the next
bits = 10
n = 1 << bits
rnd = random.Random(99)
x = orc.to_mont_n(b"".join(rnd.randrange(orc.R).to_bytes(32, "little") for _ in range(n)))
l1, l2 = wd.ntt_layout_split(bits, world)
assert l1 == l2
loc = wd.to_interleaved(torch.frombuffer(bytearray(x), dtype=torch.uint8), l1, rank, world).clone()
y = wd.dist_ntt(bn, wd.dist_ntt(bn, loc, bits, inverse=True), bits, odd=1)
want = orc.fft(orc.fft(x, n, 0, inverse=True), n, 1)
assert wd.from_interleaved(gather(y), l2).numpy().tobytes() == want
# whole proofs with the distributed CALC_H: window-sharded sums (SKIP_H) + the ranks' slices of h against their slices of the
# H points; t3 has an odd log2(domain) (the chain alternates the two splits), t6 an even one
import json, struct
gold = os.path.join(os.environ["WS_ROOT"], "tests", "golden")
for name in ("t3", "t6"):
    pkey = open(os.path.join(gold, "keys", name + ".pkey.bin"), "rb").read()
    wit = open(os.path.join(gold, "keys", name + ".witness.bin"), "rb").read()
    key = bn.load_key(pkey)
    dp = wd.DistProver(bn, key, pkey[struct.unpack_from("<I", pkey, 36)[0]:])
    w = torch.frombuffer(bytearray(wit), dtype=torch.uint8)
    for c in json.load(open(os.path.join(gold, "proofs.json")))[name]:
        got = dp.prove(w.data_ptr(), len(wit), r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"]))
        assert got == c["proof"], ("DistProver", name, rank)
    got = dp.prove(w.data_ptr(), len(wit))                     # rank 0 draws r, s
    r_used, s_used = bn.last_blinding()
    assert got == bn.groth16GenProof(wit, key, r=r_used, s=s_used), ("DistProver default blinding", name, rank)
    for v in (r_used, s_used):      # the reference's check of its draw (test/bn128_prover.js:65-71); here rank 0 drew for everybody
        assert 96 <= bin(int.from_bytes(v, "little"))[2:].count("0") <= 160, ("hamming weight of drawn blinding", rank)
# the same proofs from the NATIVE orchestration (wsnark_groth16_prove_dist, csrc/dist.hip): points shards of the key (1 / world
# resident per rank), row-sharded sparse products, pack / unpack kernels instead of permute().contiguous(), the transport as
# callbacks -- nothing in Python between the kernels.  Must equal the reference's proofs and DistProver's.
from wasmsnark_amd import formats
for name in ("t3", "t6"):
    pkey = open(os.path.join(gold, "keys", name + ".pkey.bin"), "rb").read()
    wit = open(os.path.join(gold, "keys", name + ".witness.bin"), "rb").read()
    sec = formats.pkey_bin_to_sections(pkey)
    if (1 << ((sec["domain"].bit_length() - 1) // 2)) < world:
        continue                                               # (t3: domain 8 only splits over two ranks)
    npv = wd.NativeDistProver(bn, sec)
    assert npv.key.shard["world"] == world and npv.key.shard["n_signals"] <= sec["n_vars"] // world + world
    w = torch.frombuffer(bytearray(wit), dtype=torch.uint8)
    for c in json.load(open(os.path.join(gold, "proofs.json")))[name]:
        got = npv.prove(w.data_ptr(), len(wit), r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"]))
        assert got == c["proof"], ("NativeDistProver", name, rank)
    got = npv.prove(w.data_ptr(), len(wit))                    # rank 0 draws r, s: every rank must assemble the same proof
    mine = torch.frombuffer(bytearray(json.dumps(got, sort_keys=True).encode().ljust(4096)), dtype=torch.uint8)
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert all(torch.equal(both[0], b) for b in both), "ranks disagree on the native proof"
    # Error agreement (the collectives pair up by call order): whatever goes wrong on ONE rank, EVERY rank returns an error and
    # nobody is left waiting in a collective.  (a) rank 0 injects r, s and rank 1 lets them be drawn; (b) the ranks inject different
    # values; (c) rank 1 is handed a witness that is too short; then a good proof again on the same prover.
    from wasmsnark_amd._lib import WsnarkError
    c0 = json.load(open(os.path.join(gold, "proofs.json")))[name][0]
    r_ok, s_ok = bytes.fromhex(c0["r"]), bytes.fromhex(c0["s"])
    for kw in (dict(r=r_ok if rank == 0 else None, s=s_ok if rank == 0 else None),
               dict(r=r_ok, s=s_ok if rank == 0 else bytes(x ^ 0x5A for x in s_ok)),
               dict(r=r_ok, s=s_ok, short=(rank == 1))):
        short = kw.pop("short", False)
        try:
            npv.prove(w.data_ptr(), len(wit) - (32 if short else 0), **kw)
            raise AssertionError("rank %d: a proof came out of a call the ranks disagreed on" % rank)
        except WsnarkError as ex:
            assert ex.code in (1, 4), ex
    assert npv.prove(w.data_ptr(), len(wit), r=r_ok, s=s_ok) == c0["proof"], ("NativeDistProver after errors", name, rank)
dist.barrier()
open(os.path.join(os.environ["WS_OUT"], "rank%d.ok" % rank), "w").write("ok")
'''


def test_dist_ntt_world2(tmp_path):
    from emul_util import emul_bn128
    emul_bn128()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, WS_ROOT=ROOT, WS_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", WSNARK_EMUL_DEVICES="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


WORKER4 = r'''
import os, sys, json
sys.path.insert(0, os.environ["WS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["WS_ROOT"], "tests"))
import torch, torch.distributed as dist
from emul_util import emul_bn128
from wasmsnark_amd import dist as wd, formats
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
bn = emul_bn128()
assert bn.device_info.endswith("device=%d" % (rank % 2)), bn.device_info     # (two emulated devices: rank 1 works on device 1)
gold = os.path.join(os.environ["WS_ROOT"], "tests", "golden")
pkey = open(os.path.join(gold, "keys", "t6.pkey.bin"), "rb").read()
wit = open(os.path.join(gold, "keys", "t6.witness.bin"), "rb").read()
sec = formats.pkey_bin_to_sections(pkey)
npv = wd.NativeDistProver(bn, sec)
sh = npv.key.shard
assert (sh["rank"], sh["world"], sh["h_interleave_log"], sh["n_hexps"]) == (rank, world, 3, sec["domain"] // world)
w = torch.frombuffer(bytearray(wit), dtype=torch.uint8)
for c in json.load(open(os.path.join(gold, "proofs.json")))["t6"]:
    assert npv.prove(w.data_ptr(), len(wit), r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"], ("native", rank)
dist.barrier()
open(os.path.join(os.environ["WS_OUT"], "rank%d.ok" % rank), "w").write("ok")
'''


def test_native_dist_prover_world4_and_world8(tmp_path):
    """wsnark_groth16_prove_dist over 4 and 8 ranks (gloo, emulator): points shards of 1/4 and 1/8 of the t6 key, the three
    exchanges of the distributed CALC_H among that many ranks, against the reference's own proofs."""
    from emul_util import emul_bn128
    emul_bn128()
    script = tmp_path / "worker4.py"
    script.write_text(WORKER4)
    for world, port in ((4, "29643"), (8, "29644")):
        out_dir = tmp_path / ("w%d" % world)
        out_dir.mkdir()
        env = dict(os.environ, WS_ROOT=ROOT, WS_OUT=str(out_dir), MASTER_ADDR="127.0.0.1", WSNARK_EMUL_DEVICES="2", OMP_NUM_THREADS="1")
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                              "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                             env=env, capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        assert all((out_dir / ("rank%d.ok" % r)).exists() for r in range(world))


def test_layout_helpers_roundtrip():
    import torch
    from wasmsnark_amd import dist as wd
    n, log_m, world = 64, 3, 4
    x = torch.arange(n * 32, dtype=torch.int64).to(torch.uint8)
    parts = torch.cat([wd.to_interleaved(x, log_m, r, world) for r in range(world)])
    assert torch.equal(wd.from_interleaved(parts, log_m), x)
    # rank 1 of 4 with m = 8 holds residues 2, 3: first row = elements 2, 10, 18, ...
    row0 = wd.to_interleaved(x, log_m, 1, world)[:32 * 8].view(8, 32)
    assert torch.equal(row0[1], x.view(n, 32)[10])
