// node tests/node_key_file_check.js <key file> <witness.bin> <expected proof json> <r hex> <s hex> [devices, e.g. 0,0]
// Proves from a key FILE through the Node drop-in (loadKey(path): the key never exists as a JS buffer) and compares with the expected proof.
const fs = require("fs"), path = require("path");
const ws = require(path.join(__dirname, "..", "wasmsnark_amd", "js", "index.js"));
(async () => {
    const [keyPath, witPath, wantPath, rHex, sHex, devs] = process.argv.slice(2);
    const bn = devs ? await ws.buildBn128({ devices: devs.split(",").map((x) => parseInt(x, 10)) }) : await ws.buildBn128();
    const info = bn.keyFileInfo(keyPath);
    const wit = fs.readFileSync(witPath);
    const want = JSON.parse(fs.readFileSync(wantPath, "utf8"));
    const t0 = process.hrtime.bigint();
    const h = await bn.loadKey(keyPath);
    const t1 = process.hrtime.bigint();
    const opts = { r: Buffer.from(rHex, "hex"), s: Buffer.from(sHex, "hex") };
    const got = await bn.groth16GenProof(wit, h, opts);
    const t2 = process.hrtime.bigint();
    await bn.waitTables(keyPath);
    const ts = [];
    for (let i = 0; i < 3; i++) { const a = process.hrtime.bigint(); await bn.groth16GenProof(wit, keyPath, opts); ts.push(Number(process.hrtime.bigint() - a) / 1e6); }
    const ok = JSON.stringify(got) === JSON.stringify(want);
    console.log(JSON.stringify({ ok, info, load_ms: Number(t1 - t0) / 1e6, first_proof_ms: Number(t2 - t1) / 1e6, prove_ms: ts, devices: devs || null,
        rss_MB: Math.round(process.memoryUsage().rss / 1e6) }));
    bn.terminate();
    process.exit(ok ? 0 : 1);
})().catch((e) => { console.error("NODE_KEY_FILE_FAIL", e); process.exit(1); });
