"""Child process of tests/test_gpu_key_file.py: loads ONE shard of a key file and reports how far the process's resident set rose
while it did (VmHWM after - VmRSS before), in bytes.  A fresh process: the peak is this load's, not the key generator's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def status(field):
    for line in open("/proc/self/status"):
        if line.startswith(field + ":"):
            return int(line.split()[1]) * 1024
    return 0


path, rank, world, hlog = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
import torch  # noqa: E402,F401
import wasmsnark_amd  # noqa: E402

bn = wasmsnark_amd.build(device=0)
warm = bn.load_key(path=sys.argv[5]) if len(sys.argv) > 5 else None      # (a small key first: staging ring, code objects, allocator pools)
if warm is not None:
    warm.free()
before = status("VmRSS")
key = bn.load_key(path=path, shard=(rank, world), h_interleave_log=hlog, wait_tables=False)
peak = status("VmHWM")
print(json.dumps({"rss_before": before, "hwm_after": peak, "rise": max(0, peak - before), "file_bytes": os.path.getsize(path),
                  "shard": key.shard, "load_ms": key.load_ms}))
