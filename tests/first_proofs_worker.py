"""Run by tests/test_gpu_config5.py in a fresh process: two host threads fire the FIRST two proofs of a fresh context at
the same instant on one key handle (two lanes), at a domain whose transform needs the lazily built full twiddle tables.
Both must equal the toxic-waste closed form."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    logd = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    import torch  # noqa: F401  (PyTorch's HIP runtime first, like every host of the library in this repo)
    import wasmsnark_amd
    from wasmsnark_amd import synth
    bn = wasmsnark_amd.build(device=0)
    circ = synth.NativeCircuit(bn.lib, logd, n_public=3, seed=5)
    sec, _ = circ.build_sections()
    key = bn.load_key(sections=sec)
    wit = circ.witness_bin()
    rs = [(os.urandom(32), os.urandom(32)) for _ in range(2)]
    want = [circ.expected_proof(r, s) for r, s in rs]
    got = [None, None]
    gate = threading.Barrier(2)

    def run(i):
        gate.wait()
        got[i] = bn.groth16GenProof(wit, key, r=rs[i][0], s=rs[i][1])

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got[0] == want[0] and got[1] == want[1], "a first proof differs from the closed form"
    print("first proofs OK")


if __name__ == "__main__":
    main()
