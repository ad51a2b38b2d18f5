import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from wasmsnark_amd import dist as wdist
dev = torch.device("cuda", 0)
p = bytes(range(96))
assert wdist.allgather_partials(p, dev) == p
t0 = time.perf_counter()
for _ in range(200): wdist.allgather_partials(p, dev)
print("allgather_partials (nccl, world 1): %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
dist.destroy_process_group()
