/*
 * Drop-in for wasmsnark's BN128 prover API on an MI355X (see ../../INTEGRATION.md).
 *
 * Same names and shapes as the reference:
 *   buildBn128() -> Promise<Bn128>                          (reference index.js:21, src/bn128.js:173-265)
 *   Bn128.groth16GenProof(signals, pkey) -> Promise<{pi_a, pi_b, pi_c}> of decimal strings (src/bn128.js:580-720)
 *   Bn128.g1_multiexp / g2_multiexp / calcH / terminate      (src/bn128.js:353-415, 569-578, 562-566)
 *   groth16GenProof(witness, provingKey[, cb]) and the README name genZKSnarkProof (main_bn128.js:26-39, README.md:28-30)
 * Inputs may be ArrayBuffer (what the reference requires), Buffer or TypedArray.
 * All arithmetic runs in libwsnark.so (hand-written HIP); this file only marshals and formats.
 */
"use strict";
const addon = require("./build/wsnark_napi.node");

function le2dec(u8, off) {          // bin2int of src/bn128.js:319-327
    let v = 0n;
    for (let i = 31; i >= 0; i--) v = (v << 8n) | BigInt(u8[off + i]);
    return v.toString();
}
function proofFromBytes(ab) {       // bin2g1 / bin2g2, src/bn128.js:329-351, 714-718
    const b = new Uint8Array(ab);
    const v = [];
    for (let i = 0; i < 12; i++) v.push(le2dec(b, i * 32));
    return { pi_a: [v[0], v[1], v[2]], pi_b: [[v[3], v[4]], [v[5], v[6]], [v[7], v[8]]], pi_c: [v[9], v[10], v[11]] };
}

/* digest of a key buffer -- ALL of its bytes (addon.hashBytes, off the event loop; < 0.1 s for a 0.6 GB key): enough to
 * notice that a cached handle no longer describes the bytes the caller is holding, wherever they were rewritten.  The
 * reference re-parses pkey inside every call (src/bn128.js:581-604); a caller that wants no per-call pass over the key at
 * all holds the handle of loadKey() and passes that instead of the bytes. */
async function digest(u8) {
    return Buffer.from(await addon.hashBytes(u8)).toString("hex");
}
function asBytes(x) {
    if (x instanceof ArrayBuffer) return new Uint8Array(x);
    if (ArrayBuffer.isView(x)) return new Uint8Array(x.buffer, x.byteOffset, x.byteLength);
    throw new TypeError("expected an ArrayBuffer, Buffer or TypedArray");
}

class Bn128 {
    constructor(deviceInfo) {
        this.deviceInfo = deviceInfo;
        // proving-key OBJECT (the exact ArrayBuffer / view the caller passed) -> {handle, byteOffset, byteLength, fp}.
        // Two views of one ArrayBuffer (sub-arrays of a bundle, Node's pooled small Buffers) are different keys here,
        // and a cached handle is only reused while offset, length and the digest of the WHOLE buffer still match; the
        // reference re-reads pkey on every call (src/bn128.js:581-604) -- callers that want no check at all hold the
        // handle of loadKey().
        this._keys = new WeakMap();
        this._pr = null;     // blinding values of the last proof, "for tests" like the reference (src/bn128.js:662-664)
        this._ps = null;
        this._live = true;
    }
    g1_multiexp(scalars, points) { return addon.g1Multiexp(scalars, points); }
    g2_multiexp(scalars, points) { return addon.g2Multiexp(scalars, points); }
    calcH(signals, polsA, polsB, nSignals, domainSize) { return addon.calcH(signals, polsA, polsB, nSignals, domainSize); }
    fft(buf, odd) { return addon.fft(buf, odd | 0, false); }
    ifft(buf, odd) { return addon.fft(buf, odd | 0, true); }
    /* pkey bytes -> device-resident key handle (stays in HBM across proofs; freed by the GC) */
    async loadKey(pkey) {
        if (pkey !== null && typeof pkey === "object" && !(pkey instanceof ArrayBuffer) && !ArrayBuffer.isView(pkey)) return pkey;   // already a handle
        const u8 = asBytes(pkey);
        const fp = await digest(u8);
        const hit = this._keys.get(pkey);
        if (hit && hit.byteOffset === u8.byteOffset && hit.byteLength === u8.byteLength && hit.fp === fp) return hit.handle;
        const handle = await addon.loadKey(pkey);
        this._keys.set(pkey, { handle, byteOffset: u8.byteOffset, byteLength: u8.byteLength, fp });
        return handle;
    }
    /* pkey: proving_key.bin bytes, or a handle from loadKey().
     * opts.r / opts.s: optional 32-byte blinding values (the reference draws them from crypto.randomBytes) */
    async groth16GenProof(signals, pkey, opts) {
        const h = await this.loadKey(pkey);
        const out = await addon.prove(h, signals, opts && opts.r ? opts.r : null, opts && opts.s ? opts.s : null);
        this._pr = new Uint8Array(out.slice(384, 416));
        this._ps = new Uint8Array(out.slice(416, 448));
        return proofFromBytes(out.slice(0, 384));
    }
    /* src/bn128.js:722-791: verificationKey = snarkjs "groth" verification_key.json object, input = public signals
     * (one value is wrapped like the reference does), proof = {pi_a, pi_b, pi_c}.  Native host arithmetic, no GPU. */
    async groth16Verify(verificationKey, input, proof) {
        if (input === undefined || input === null) input = [];
        else if (!Array.isArray(input)) input = [input];
        const vals = input.map((x) => BigInt(x));
        if (vals.some((v) => v < 0n || v >= (1n << 256n))) return false;
        const IC = verificationKey.IC;
        if (!IC || IC.length < vals.length + 1) throw new Error("verification key has fewer IC points than inputs + 1");
        const g1 = (p) => [p[0], p[1]], g2 = (p) => [p[0][0], p[0][1], p[1][0], p[1][1]];
        const vk = le32cat([].concat(g1(verificationKey.vk_alfa_1), g2(verificationKey.vk_beta_2), g2(verificationKey.vk_gamma_2),
            g2(verificationKey.vk_delta_2), ...IC.slice(0, vals.length + 1).map(g1)));
        const pf = le32cat([].concat(proof.pi_a, proof.pi_b[0], proof.pi_b[1], proof.pi_b[2], proof.pi_c));
        return addon.verify(vk, le32cat(vals), pf);
    }
    /* src/bn128.js:562-566.  The GPU context is process-wide (one per addon): it is shut down when the LAST live Bn128
     * object terminates, so one object's terminate() does not pull the device out from under another's proofs. */
    terminate() {
        if (!this._live) return;
        this._live = false;
        if (--liveInstances === 0) addon.shutdown();
    }
}
let liveInstances = 0;
function le32cat(list) {            // decimal strings / BigInts -> concatenated 32-byte little-endian integers
    const out = new Uint8Array(32 * list.length);
    list.forEach((x, k) => { let v = BigInt(x); for (let i = 0; i < 32; i++) { out[32 * k + i] = Number(v & 0xffn); v >>= 8n; } });
    return out;
}

let singleton = null;
/* opts.lib: load this build of the C ABI instead of the in-tree libwsnark.so.  The test-suite passes the CPU
 * thread-emulator build; nothing else should: there is no CPU path in the product. */
async function buildBn128(device, opts) {
    const info = addon.init(device === undefined || device === null ? -1 : device, opts && opts.lib ? String(opts.lib) : undefined);
    liveInstances++;
    return new Bn128(info);
}
function groth16GenProof(witness, provingKey, cb) {   // main_bn128.js:26-39
    const p = (async () => {
        if (!singleton) singleton = await buildBn128();
        return singleton.groth16GenProof(witness, provingKey);
    })();
    if (cb) { p.then((proof) => cb(null, proof), (err) => cb(err)); return undefined; }
    return p;
}

function groth16Verify(verificationKey, input, proof, cb) {   // main_bn128.js:41-55
    const p = (async () => {
        if (!singleton) singleton = await buildBn128();
        return singleton.groth16Verify(verificationKey, input, proof);
    })();
    if (cb) { p.then((ok) => cb(null, ok), (err) => cb(err)); return undefined; }
    return p;
}

const formats = require("./formats.js");     // snarkjs JSON -> proving_key.bin / witness.bin (reference tools/build*.js)
module.exports = { buildBn128, groth16GenProof, genZKSnarkProof: groth16GenProof, groth16Verify, Bn128, proofFromBytes,
    pkeyJsonToBin: formats.pkeyJsonToBin, witnessJsonToBin: formats.witnessJsonToBin };
