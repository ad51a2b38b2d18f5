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

class Bn128 {
    constructor(deviceInfo) {
        this.deviceInfo = deviceInfo;
        this._keys = new WeakMap();  // proving-key buffer -> device-resident handle (stays in HBM across proofs)
    }
    g1_multiexp(scalars, points) { return addon.g1Multiexp(scalars, points); }
    g2_multiexp(scalars, points) { return addon.g2Multiexp(scalars, points); }
    calcH(signals, polsA, polsB, nSignals, domainSize) { return addon.calcH(signals, polsA, polsB, nSignals, domainSize); }
    fft(buf, odd) { return addon.fft(buf, odd | 0, false); }
    ifft(buf, odd) { return addon.fft(buf, odd | 0, true); }
    async loadKey(pkey) {
        const id = (pkey && pkey.buffer instanceof ArrayBuffer && !(pkey instanceof ArrayBuffer)) ? pkey.buffer : pkey;
        let h = (typeof id === "object" && id !== null) ? this._keys.get(id) : undefined;
        if (!h) {
            h = await addon.loadKey(pkey);
            if (typeof id === "object" && id !== null) this._keys.set(id, h);
        }
        return h;
    }
    /* opts.r / opts.s: optional 32-byte blinding values (the reference draws them from crypto.randomBytes) */
    async groth16GenProof(signals, pkey, opts) {
        const h = await this.loadKey(pkey);
        const out = await addon.prove(h, signals, opts && opts.r ? opts.r : null, opts && opts.s ? opts.s : null);
        return proofFromBytes(out);
    }
    terminate() { addon.shutdown(); }
}

let singleton = null;
async function buildBn128(device) {
    const info = addon.init(device === undefined ? -1 : device);
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

const formats = require("./formats.js");     // snarkjs JSON -> proving_key.bin / witness.bin (reference tools/build*.js)
module.exports = { buildBn128, groth16GenProof, genZKSnarkProof: groth16GenProof, Bn128, proofFromBytes,
    pkeyJsonToBin: formats.pkeyJsonToBin, witnessJsonToBin: formats.witnessJsonToBin };
