import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections(); key = bn.load_key(sections=sec); wit = circ.witness_bin()
r, s = bytes(range(32)), bytes(range(32, 64)); want = circ.expected_proof(r, s)
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
pinned = torch.frombuffer(bytearray(wit), dtype=torch.uint8).pin_memory(); torch.cuda.synchronize()
def t(f, n=20):
    for _ in range(3): out = f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
base, _ = t(lambda: bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s))
print(json.dumps({"resident": round(base, 3)}), flush=True)
def copy_then_prove():
    d_w.copy_(pinned, non_blocking=True); torch.cuda.synchronize()
    return bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s)
ms, p = t(copy_then_prove); print(json.dumps({"torch copy + sync, then resident proof": round(ms, 3), "over": round(ms - base, 3), "ok": p == want}), flush=True)
for kb in (0, 2048, 4096, 8192, 16384, 32768, 65536):
    for chunked in (1, 0):
        bn.lib.tune("STAGE_DIRECT_CHUNK_KB", kb or None); bn.lib.tune("PROVE_CHUNKED_UPLOAD", chunked)
        ms, p = t(lambda: bn.groth16GenProof_hostptr(pinned.data_ptr(), len(wit), key, r=r, s=s))
        print(json.dumps({"pinned source, DMA chunk KiB": kb or "default 4096", "chunked_histogram": chunked, "ms": round(ms, 3), "over_resident": round(ms - base, 3), "ok": p == want}), flush=True)
bn.lib.tune("STAGE_DIRECT_CHUNK_KB", None)
for kb in (4096, 8192, 16384):
    for w in (4, 8):
        bn.lib.tune("STAGE_CHUNK_KB", kb); bn.lib.tune("STAGE_WORKERS", w); bn.lib.tune("PROVE_CHUNKED_UPLOAD", 1)
        ms, p = t(lambda: bn.groth16GenProof(wit, key, r=r, s=s))
        print(json.dumps({"pageable source, chunk KiB": kb, "workers": w, "ms": round(ms, 3), "over_resident": round(ms - base, 3), "ok": p == want}), flush=True)
