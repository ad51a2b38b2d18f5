#!/bin/bash
# Samples the engine clock and the socket power while the GPU runs (a) the class probes' plain-add and multiply-add loops, (b) whole proofs.
# Why: the issue-class rates of tools/issue_probe.py are lane-ops per SECOND; turning them into cycles needs the clock each load runs at,
# and a power-managed part does not run every instruction mix at one clock.   Usage (GPU box): bash tools/clock_probe.sh > out.txt
cd "${GRAFT_REPO_ROOT:-$(dirname $0)/..}"
sample() {   # $1 = label, $2 = seconds
    for i in $(seq 1 $2); do
        c=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1 | sed 's/.*(\(.*\)).*/\1/')
        p=$(rocm-smi --showpower 2>/dev/null | grep -i "power" | grep -v "===" | head -1 | awk -F: '{print $NF}')
        t=$(rocm-smi --showtemp 2>/dev/null | grep -i "junction\|hotspot" | head -1 | awk -F: '{print $NF}')
        echo "$1 sclk=$c power_W=$p temp_C=$t"
        sleep 1
    done
}
echo "== idle"; sample idle 3
python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, wasmsnark_amd
from wasmsnark_amd import synth
bn = wasmsnark_amd.build(device=0)
r, s = bytes(range(32)), bytes(range(32, 64))
circ = synth.NativeCircuit(bn.lib, 20, n_public=5, seed=1)
sec, _ = circ.build_sections(); wit = circ.witness_bin()
d_w = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda(); torch.cuda.synchronize()
key = bn.load_key(sections=sec)
bn.lib.c.wsnark_pkey_wait_tables(key._h) if hasattr(key, "_h") else None
open("/tmp/clock_probe_ready", "w").write("1")
t0 = time.time(); n = 0
while time.time() - t0 < 14:
    bn.groth16GenProof_dev(d_w.data_ptr(), len(wit), key, r=r, s=s); n += 1
print("proofs", n, "ms per proof", round((time.time() - t0) / n * 1e3, 3), flush=True)
PY
while [ ! -f /tmp/clock_probe_ready ]; do sleep 0.5; done
echo "== proofs back to back"; sample proofs 10
wait
rm -f /tmp/clock_probe_ready
python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import wasmsnark_amd
import ctypes as C
bn = wasmsnark_amd.build(device=0)
open("/tmp/clock_probe_ready", "w").write("1")
for what, label in ((6, "v_add_u32 probe"), (10, "v_mad_u64_u32 probe"), (1, "product chain probe"), (303, "register-resident G1 addition loop, 3 wavefronts per SIMD")):
    t0 = time.time(); best = 0.0
    while time.time() - t0 < 6:
        v = C.c_double(0)
        bn.lib.check(bn.lib.c.wsnark_peak_probe(what, C.byref(v))); best = max(best, v.value)
    print("done", label, "best", round(best, 1), "at t =", round(time.time(), 1), flush=True)
PY
while [ ! -f /tmp/clock_probe_ready ]; do sleep 0.5; done
echo "== probes: 6 s each of v_add_u32, v_mad_u64_u32, the product chain, the register-resident G1 addition loop"; date +%s.%N; sample probes 25
wait
rm -f /tmp/clock_probe_ready
