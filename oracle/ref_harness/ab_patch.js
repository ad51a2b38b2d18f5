/*
 * ab_patch.js -- boundary A/B in the reference's OWN caller (INTEGRATION.md section 1).  TEST INFRASTRUCTURE ONLY.
 *
 * Loads the read-only reference Bn128 (by path, through refenv.js), replaces its three seam methods
 *     Bn128.g1_multiexp / g2_multiexp / calcH          (src/bn128.js:353-415, 569-578)
 * by the build's N-API addon (wasmsnark_amd/js), and then runs the reference's UNMODIFIED groth16GenProof
 * (src/bn128.js:580-720: key slicing, blinding, the scalar multiplications, affine + decimal output) and its
 * UNMODIFIED groth16Verify (:722-791, the WASM pairing) on the committed keys.  The patched prover must give the same
 * proofs as the stock reference (tests/golden/proofs.json) and the reference verifier must accept them.
 *
 * In the build container there is no GPU: the addon is pointed at the CPU thread-emulator build of the SAME kernel
 * sources (tests/emul/libwsnark_emul.so), explicitly.  The result is written to tests/golden/ab_patch.json, which is
 * what travels; tests/test_ab_patch.py re-runs this script when /root/reference is present.
 *
 *   node oracle/ref_harness/ab_patch.js <path of the libwsnark build to load> [--write]
 */
"use strict";
const fs = require("fs");
const path = require("path");
const E = require("./refenv.js");
const { hex, le32, toAB } = E;

const ROOT = path.join(__dirname, "..", "..");
const OUT = path.join(ROOT, "tests", "golden");

(async () => {
    // argv[2] == "emul" (what tests/test_ab_patch.py passes in the build container, which has no GPU): the emulator build of the addon
    if (process.argv[2] === "emul" || /emul/.test(process.argv[2] || "")) require(path.join(ROOT, "tests", "emul", "use_emulator_addon.js"));
    const { bn } = await E.buildRef();
    const ws = require(path.join(ROOT, "wasmsnark_amd", "js", "index.js"));
    const mine = await ws.buildBn128();
    const calls = { g1_multiexp: 0, g2_multiexp: 0, calcH: 0 };
    // the monkey-patch of INTEGRATION.md section 1: same names, same arguments, same result bytes
    bn.g1_multiexp = (scalars, points) => { calls.g1_multiexp++; return mine.g1_multiexp(scalars, points); };
    bn.g2_multiexp = (scalars, points) => { calls.g2_multiexp++; return mine.g2_multiexp(scalars, points); };
    bn.calcH = (signals, polsA, polsB, nSignals, domainSize) => { calls.calcH++; return mine.calcH(signals, polsA, polsB, nSignals, domainSize); };

    const stock = JSON.parse(fs.readFileSync(path.join(OUT, "proofs.json"), "utf8"));
    const dir = path.join(OUT, "keys");
    const out = { how: "reference Bn128 with g1_multiexp / g2_multiexp / calcH replaced by the build's N-API addon; reference groth16GenProof + groth16Verify unmodified",
                  addon_device: mine.deviceInfo, cases: {} };
    let all = true;
    for (const name of Object.keys(stock).sort()) {
        const pkey = fs.readFileSync(path.join(dir, name + ".pkey.bin"));
        const wit = fs.readFileSync(path.join(dir, name + ".witness.bin"));
        const vk = JSON.parse(fs.readFileSync(path.join(dir, name + ".vk.json"), "utf8"));
        const pub = JSON.parse(fs.readFileSync(path.join(dir, name + ".public.json"), "utf8"));
        if (!vk.vk_alfabeta_12) vk.vk_alfabeta_12 = [[["0", "0"], ["0", "0"], ["0", "0"]], [["0", "0"], ["0", "0"], ["0", "0"]]];
        out.cases[name] = [];
        for (const c of stock[name]) {
            E.setRS(Uint8Array.from(Buffer.from(c.r, "hex")), Uint8Array.from(Buffer.from(c.s, "hex")));
            const proof = await bn.groth16GenProof(toAB(new Uint8Array(wit)), toAB(new Uint8Array(pkey)));
            const same = JSON.stringify(proof) === JSON.stringify(c.proof);
            const ok = await bn.groth16Verify(vk, pub, proof);
            const bad = pub.length ? await bn.groth16Verify(vk, [String(BigInt(pub[0]) + 1n)].concat(pub.slice(1)), proof) : null;
            out.cases[name].push({ r: c.r, s: c.s, proof, same_as_stock_reference: same, reference_verifies: ok,
                                   reference_rejects_wrong_public: bad === null ? null : !bad });
            all = all && same && ok && (bad === null || !bad);
            console.log(name, "r=" + c.r.slice(0, 8), "same as stock:", same, "verify:", ok, "wrong-public rejected:", bad === null ? "n/a" : !bad);
        }
    }
    out.seam_calls = calls;
    if (process.argv.includes("--write")) fs.writeFileSync(path.join(OUT, "ab_patch.json"), JSON.stringify(out, null, 0));
    console.log(all && calls.calcH > 0 && calls.g2_multiexp > 0 ? "AB_PATCH_OK" : "AB_PATCH_FAIL", JSON.stringify(calls));
    mine.terminate();
    bn.terminate();
})().catch((e) => { console.error("AB_PATCH_FAIL", e); process.exit(1); });
