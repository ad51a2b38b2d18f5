#!/usr/bin/env python3
"""Proof from a HOST witness against the resident-witness call, one process, the staging settings swept through the library's
A/B switches (wsnark_tuning_set): workers x chunk size, chunked histogram on / off, and a PINNED source (torch pinned tensor:
DMA in place, no staging).  Prints one JSON line per setting.   python tools/upload_sweep.py [log_domain] [reps]"""
import json
import os
import sys
import time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import wasmsnark_amd
from wasmsnark_amd import synth
logd = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, logd, n_public=5, seed=1)
sec, _ = circ.build_sections()
key = bn.load_key(sections=sec)
wit = circ.witness_bin()
r, s = bytes(range(32)), bytes(range(32, 64))
want = circ.expected_proof(r, s)
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
pinned = torch.frombuffer(bytearray(wit), dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()


def t(f, n=reps):
    for _ in range(3):
        out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


dev = lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
host = lambda: bn.groth16GenProof(wit, key, r=r, s=s)
pin = lambda: bn.groth16GenProof_hostptr(pinned.data_ptr(), len(wit), key, r=r, s=s)
base, p = t(dev)
print(json.dumps({"setting": "resident witness", "ms": round(base, 3), "ok": p == want}), flush=True)
# the transfer itself: 32 MiB from pinned memory with nothing else on the GPU -- one copy on one queue, and halves on two queues
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
dst = torch.empty_like(d_w)
half = pinned.numel() // 2


def copy_one():
    with torch.cuda.stream(s1):
        dst.copy_(pinned, non_blocking=True)
    s1.synchronize()


def copy_two():
    with torch.cuda.stream(s1):
        dst[:half].copy_(pinned[:half], non_blocking=True)
    with torch.cuda.stream(s2):
        dst[half:].copy_(pinned[half:], non_blocking=True)
    s1.synchronize(); s2.synchronize()


for name, f in (("one DMA, one queue", copy_one), ("two halves on two queues", copy_two)):
    ms, _ = t(f)
    print(json.dumps({"setting": "raw H2D of the %d-byte witness from pinned memory, %s" % (len(wit), name), "ms": round(ms, 3), "GBps": round(len(wit) / ms / 1e6, 1)}), flush=True)
for dual in (1, 0):
    for chunked in (1, 0):
        for workers in (0, 2, 4, 8):
            for chunk_kb in (0, 4096, 16384):
                if (workers == 0) != (chunk_kb == 0):
                    continue
                bn.lib.tune("STAGE_DUAL", dual); bn.lib.tune("PROVE_CHUNKED_UPLOAD", chunked)
                bn.lib.tune("STAGE_WORKERS", workers or None); bn.lib.tune("STAGE_CHUNK_KB", chunk_kb or None)
                ms, p = t(host)
                print(json.dumps({"setting": {"two_copy_queues": dual, "chunked_histogram": chunked, "workers": workers or "default (4)", "chunk_kb": chunk_kb or "default (8192)"},
                                  "host_witness_ms": round(ms, 3), "over_resident_ms": round(ms - base, 3), "ok": p == want}), flush=True)
for name in ("PROVE_CHUNKED_UPLOAD", "STAGE_WORKERS", "STAGE_CHUNK_KB"):
    bn.lib.tune(name, None)
for dual in (1, 0):
    bn.lib.tune("STAGE_DUAL", dual)
    ms, p = t(pin)
    print(json.dumps({"setting": "pinned source (torch pin_memory), DMA in place, two copy queues = %d" % dual, "host_witness_ms": round(ms, 3), "over_resident_ms": round(ms - base, 3), "ok": p == want}), flush=True)
bn.lib.tune("STAGE_DUAL", None)
base2, _ = t(dev)
print(json.dumps({"setting": "resident witness (again)", "ms": round(base2, 3)}), flush=True)
