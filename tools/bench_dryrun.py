"""CPU dry run of bench.py's control flow against the thread-emulator build (tiny sizes): catches Python-side mistakes
before GPU minutes are spent.  Not a measurement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *_: None
torch.cuda.synchronize = lambda *_: None
_real_device = torch.device
torch.device = lambda *a, **k: _real_device("cpu")
import torch.distributed as _dist

_real_init = _dist.init_process_group
_dist.init_process_group = lambda backend=None, **kw: _real_init("gloo")      # N > 1 dry run: gloo instead of nccl
import __graft_entry__

__graft_entry__.ensure_built = lambda: None
import wasmsnark_amd
from emul_util import emul_bn128

wasmsnark_amd.build = lambda device=-1: emul_bn128()
import bench

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
