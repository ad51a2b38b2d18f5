"""-m gpu tests of the N > 1 code.  Two of them need MORE THAN ONE visible GPU and skip themselves otherwise (the build's gpurun
boxes have one; the driver's 8-GPU node, when there is one, runs them).  The third runs everywhere: TWO ranks sharing ONE
GPU over gloo (RCCL refuses two ranks on one device) -- the real kernels, real asynchronous streams and a real exchange
between two processes, only the transport differs from the multi-GPU case.  One process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI), the same code path bench.py --gpus N uses: window-sharded MSM and whole sharded proofs against
the reference's golden proofs, and an entry point called from a thread other than the one that initialised a
context on device 1."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


WORKER = r'''
import json, os, random, sys
sys.path.insert(0, os.environ["WS_ROOT"])
import torch, torch.distributed as dist
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
import wasmsnark_amd
from wasmsnark_amd import dist as wd
from oracle import pyoracle as orc
bn = wasmsnark_amd.build(device=local)
rnd = random.Random(42)
for g, n in ((1, 3000), (2, 700)):
    sz = 64 if g == 1 else 128
    pts = bn.mul_base(g, b"".join(rnd.randrange(1, orc.R).to_bytes(32, "little") for _ in range(n)))
    sc = b"".join(rnd.randrange(1 << 256).to_bytes(32, "little") for _ in range(n))
    want = orc.g_affine(g, orc.multiexp(g, "workers8", sc, pts, n))
    f = bn.g1_multiexp if g == 1 else bn.g2_multiexp
    lo, hi = wd.shard_bounds(n, world, rank)                      # the reference's split (src/bn128.js:353-383)
    assert wd.sharded_msm(bn, g, f(sc[lo * 32:hi * 32], pts[lo * sz:hi * sz]), dev) == want, ("points", g, rank)
    assert wd.sharded_msm(bn, g, f(sc, pts, shard=(rank, world)), dev) == want, ("windows", g, rank)
gold = os.path.join(os.environ["WS_ROOT"], "tests", "golden")
key = bn.load_key(open(os.path.join(gold, "keys", "t6.pkey.bin"), "rb").read())
wit = open(os.path.join(gold, "keys", "t6.witness.bin"), "rb").read()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).to(dev)
torch.cuda.synchronize()
for c in json.load(open(os.path.join(gold, "proofs.json")))["t6"]:
    r, s = bytes.fromhex(c["r"]), bytes.fromhex(c["s"])
    assert wd.sharded_prove(bn, key, wit, r=r, s=s, device=dev) == c["proof"], ("sharded prove", rank)
    assert wd.sharded_prove(bn, key, None, r=r, s=s, device=dev, d_witness=(d_w.data_ptr(), len(wit))) == c["proof"]
# distributed four-step NTT over RCCL (one all_to_all_single per transform) against the single-GPU transform
for bits in (10, 16, 20):
    n = 1 << bits
    g = torch.Generator(device="cpu").manual_seed(bits)
    x = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, generator=g)
    x[31::32] &= 0x1F
    d = x.to(dev)
    l1, l2 = wd.ntt_layout_split(bits, world)
    for odd, inverse in ((0, False), (1, False), (0, True), (1, True)):
        ref = d.clone(); torch.cuda.synchronize()
        bn.fft_dev(ref.data_ptr(), n, odd, inverse=inverse); torch.cuda.synchronize()
        y = wd.dist_ntt(bn, wd.to_interleaved(d, l1, rank, world).clone(), bits, odd=odd, inverse=inverse)
        assert torch.equal(wd.from_interleaved(wd.gather_all(y), l2), ref), ("dist_ntt", bits, odd, inverse, rank)
# whole proofs with the distributed CALC_H (DistProver): 2^14 synthetic circuit against the toxic-waste closed form
import struct
from wasmsnark_amd import synth
circ = synth.make_circuit(14, n_public=5, seed=314)
S = synth.setup(circ, seed=15)
pk14, _ = synth.build_key(circ, S, bn.mul_base)
key14 = bn.load_key(pk14)
w14 = synth.witness_bin(circ)
d_w14 = torch.frombuffer(bytearray(w14), dtype=torch.uint8).to(dev)
torch.cuda.synchronize()
dp = wd.DistProver(bn, key14, pk14[struct.unpack_from("<I", pk14, 36)[0]:], device=dev)
r14, s14 = bytes(range(32)), bytes(range(64, 96))
want14 = synth.expected_proof(circ, S, r14, s14, bn.mul_base)
assert dp.prove(d_w14.data_ptr(), len(w14), r=r14, s=s14) == want14, ("DistProver", rank)
assert wd.sharded_prove(bn, key14, w14, r=r14, s=s14, device=dev) == want14, ("sharded_prove 2^14", rank)
# the native orchestration (wsnark_groth16_prove_dist): every rank holds a POINTS shard of the key, the transport is called
# back on the library's own queue; 2^14 and 2^18 against the closed form, injected and rank-0-drawn blinding
from wasmsnark_amd import formats
npv = wd.NativeDistProver(bn, formats.pkey_bin_to_sections(pk14), device=dev)
assert npv.key.shard["world"] == world and (npv.key.table["bytes"] < key14.table["bytes"] or world == 1)
assert npv.prove(d_w14.data_ptr(), len(w14), r=r14, s=s14) == want14, ("NativeDistProver", rank)
got_n = npv.prove(d_w14.data_ptr(), len(w14))
r_n, s_n = bn.last_blinding()
assert got_n == synth.expected_proof(circ, S, r_n, s_n, bn.mul_base), ("NativeDistProver, drawn blinding", rank)
c18 = synth.NativeCircuit(bn.lib, 18, n_public=5, seed=18)
sec18, _ = c18.build_sections()
w18 = c18.witness_bin()
d_w18 = torch.frombuffer(bytearray(w18), dtype=torch.uint8).to(dev)
torch.cuda.synchronize()
np18 = wd.NativeDistProver(bn, sec18, device=dev)
for _ in range(3):
    assert np18.prove(d_w18.data_ptr(), len(w18), r=r14, s=s14) == c18.expected_proof(r14, s14), ("NativeDistProver 2^18", rank)
got = wd.sharded_prove(bn, key, wit, device=dev)                  # rank 0 draws r, s: all ranks, one proof
r_used, s_used = bn.last_blinding()
assert got == bn.groth16GenProof(wit, key, r=r_used, s=s_used)
dist.barrier()
open(os.path.join(os.environ["WS_OUT"], "rank%d.ok" % rank), "w").write("ok")
dist.destroy_process_group()
'''


def test_sharded_msm_and_prove_nccl(tmp_path):
    n = _gpus()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % n)
    world = 2
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, WS_ROOT=ROOT, WS_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert all((tmp_path / ("rank%d.ok" % r)).exists() for r in range(world))


def test_rccl_world_of_one_on_this_gpu(tmp_path):
    """The SAME worker over backend "nccl" with ONE rank: what a one-GPU box can show of the RCCL transport -- the process group
    on a device, all_gather_into_tensor / all_gather of the byte records, all_to_all_single enqueued on the LIBRARY's queue
    (torch.cuda.ExternalStream in NativeDistProver's callback) -- none of which short-circuits for a world of one
    (wasmsnark_amd/dist.py: _all_to_all, allgather_partials, gather_all).  What it cannot show is a byte crossing xGMI."""
    if _gpus() < 1:
        pytest.skip("needs a GPU")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, WS_ROOT=ROOT, WS_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29635", str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (tmp_path / "rank0.ok").exists()


THREAD_WORKER = r'''
import os, sys, threading
sys.path.insert(0, os.environ["WS_ROOT"])
import json
import wasmsnark_amd
bn = wasmsnark_amd.build(device=1)          # context on GPU 1, initialised on the main thread
gold = os.path.join(os.environ["WS_ROOT"], "tests", "golden")
key = bn.load_key(open(os.path.join(gold, "keys", "t6.pkey.bin"), "rb").read())
wit = open(os.path.join(gold, "keys", "t6.witness.bin"), "rb").read()
c = json.load(open(os.path.join(gold, "proofs.json")))["t6"][1]
res = []
def work():                                  # a fresh thread: HIP's current device defaults to 0 there
    res.append(bn.groth16GenProof(wit, key, r=bytes.fromhex(c["r"]), s=bytes.fromhex(c["s"])) == c["proof"])
    x = bn.toMontgomeryN((5).to_bytes(32, "little") * 1000)
    res.append(bn.fromMontgomeryN(x) == (5).to_bytes(32, "little") * 1000)
t = threading.Thread(target=work); t.start(); t.join()
assert res == [True, True], res
print("ok")
'''


def test_entry_points_from_another_thread_on_device_1(tmp_path):
    n = _gpus()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (found %d)" % n)
    script = tmp_path / "thread_worker.py"
    script.write_text(THREAD_WORKER)
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, WS_ROOT=ROOT), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


ONE_GPU_WORKER = WORKER.replace('dist.init_process_group("nccl", device_id=dev)', 'dist.init_process_group("gloo")') \
    .replace('rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])',
             'rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), 0')


def test_two_ranks_sharing_one_gpu_over_gloo(tmp_path):
    """The whole N > 1 surface with 2 processes on GPU 0: points split, window split, sharded_prove (host and device
    witness, injected and rank-0-drawn blinding), dist_ntt up to 2^20 against the single-GPU transform, DistProver against
    the closed form."""
    if _gpus() < 1:
        pytest.skip("needs a GPU")
    assert 'init_process_group("gloo")' in ONE_GPU_WORKER and ", 0\n" in ONE_GPU_WORKER
    script = tmp_path / "worker_one_gpu.py"
    script.write_text(ONE_GPU_WORKER)
    env = dict(os.environ, WS_ROOT=ROOT, WS_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29633", str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert all((tmp_path / ("rank%d.ok" % r)).exists() for r in range(2))
