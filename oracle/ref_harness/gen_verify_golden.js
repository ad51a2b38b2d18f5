/*
 * gen_verify_golden.js -- verdicts of the READ-ONLY reference verifier (Bn128.groth16Verify, src/bn128.js:722-791, loaded
 * by path through refenv.js) on the verifier data the reference's own tests / example hold:
 *     example/bn128/verification_key.json (= test/data/verification_key.json), public.json, proof.json,
 *     proof_good.json, proof_good0.json                      (test/bn128_prover.js:66-80, example/bn128/index.html)
 * plus tampered variants (a public input changed, an input >= r, pi_c swapped for pi_a, pi_b negated).
 * Writes tests/golden/verify.json = the inputs (DATA: key, public signals, proofs) and the reference's verdicts.
 * TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.
 */
"use strict";
const fs = require("fs");
const path = require("path");
const E = require("./refenv.js");
const OUT = path.join(__dirname, "..", "..", "tests", "golden");
const Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;

(async () => {
    const { bn } = await E.buildRef();
    const ex = path.join(E.REF, "example", "bn128");
    const rd = (f) => JSON.parse(fs.readFileSync(path.join(ex, f), "utf8"));
    const vk = rd("verification_key.json"), pub = rd("public.json");
    const proofs = { "proof.json": rd("proof.json"), "proof_good.json": rd("proof_good.json"), "proof_good0.json": rd("proof_good0.json") };
    const cases = [];
    const run = async (label, proofName, proof, inputs) => {
        const ok = await bn.groth16Verify(vk, inputs, proof);
        cases.push({ label, proof_file: proofName, proof, inputs, reference_verdict: !!ok });
        console.log(label, proofName, "->", ok);
    };
    for (const [name, proof] of Object.entries(proofs)) {
        await run("as shipped", name, proof, pub);
        const bad = pub.slice(); bad[0] = String((BigInt(bad[0]) + 1n) % R);
        await run("first public input + 1", name, proof, bad);
        const last = pub.slice(); last[last.length - 1] = String((BigInt(last[last.length - 1]) + 5n) % R);
        await run("last public input + 5", name, proof, last);
        const big = pub.slice(); big[3] = String(BigInt(big[3]) + R);
        await run("an input >= r (same residue)", name, proof, big);
        await run("pi_c replaced by pi_a", name, { pi_a: proof.pi_a, pi_b: proof.pi_b, pi_c: proof.pi_a, protocol: proof.protocol }, pub);
        const nb = [proof.pi_b[0], [String((Q - BigInt(proof.pi_b[1][0])) % Q), String((Q - BigInt(proof.pi_b[1][1])) % Q)], proof.pi_b[2]];
        await run("pi_b negated", name, { pi_a: proof.pi_a, pi_b: nb, pi_c: proof.pi_c, protocol: proof.protocol }, pub);
    }
    const { vk_alfabeta_12, ...vkSmall } = vk;        // the 12 Fq12 coefficients are not an input of the check that is run (:783)
    fs.writeFileSync(path.join(OUT, "verify.json"), JSON.stringify({ source: "reference example/bn128 + test/data verifier files; verdicts from the reference's groth16Verify",
        verification_key: vkSmall, cases }, null, 0));
    bn.terminate();
})().catch((e) => { console.error(e); process.exit(1); });
