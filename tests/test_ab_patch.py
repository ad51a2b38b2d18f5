"""Boundary A/B in the reference's own caller: the reference Bn128 with its three seam methods
g1_multiexp / g2_multiexp / calcH (src/bn128.js:353-415, 569-578) replaced by the build's N-API addon, then the
reference's UNMODIFIED groth16GenProof (:580-720) and groth16Verify (:722-791).
The committed fixture tests/golden/ab_patch.json is the outcome of oracle/ref_harness/ab_patch.js in the build container
(addon on the CPU thread-emulator build of the kernel sources); where /root/reference and node exist the run is repeated."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT, load_golden

REF = os.environ.get("WSNARK_REF", "/root/reference")


def test_committed_fixture_says_identical_and_verified():
    ab, stock = load_golden("ab_patch.json"), load_golden("proofs.json")
    assert sorted(ab["cases"]) == sorted(stock)
    for name, cases in ab["cases"].items():
        assert len(cases) == len(stock[name])
        for got, want in zip(cases, stock[name]):
            assert got["proof"] == want["proof"] and (got["r"], got["s"]) == (want["r"], want["s"])
            assert got["same_as_stock_reference"] and got["reference_verifies"] and got["reference_rejects_wrong_public"]
    calls = ab["seam_calls"]
    n = sum(len(c) for c in stock.values())
    assert calls == {"g1_multiexp": 4 * n, "g2_multiexp": n, "calcH": n}     # src/bn128.js:607-620: 4 + 1 + 1 per proof


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")) or shutil.which("node") is None
                    or not os.path.exists("/usr/include/node/node_api.h"),
                    reason="needs the reference checkout and node (build container only)")
def test_rerun_in_the_reference_caller(tmp_path):
    from emul_util import emul_bn128, SO
    emul_bn128()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "wasmsnark_amd", "js"), "-s", "all", "emul"])
    out = subprocess.run(["node", os.path.join(ROOT, "oracle", "ref_harness", "ab_patch.js"), "emul"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "AB_PATCH_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
