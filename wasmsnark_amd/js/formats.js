/*
 * snarkjs JSON -> wasmsnark binary inputs, the Node side of wasmsnark_amd/formats.py.
 *
 *   pkeyJsonToBin(obj) -> Buffer      replaces tools/buildpkey.js:124-186 of the reference
 *   witnessJsonToBin(arr) -> Buffer   replaces tools/buildwitness.js:36-69
 *
 * Native BigInt, no dependencies.  Values are decimal strings (or numbers / BigInts); field elements go to
 * Montgomery form (x * 2^256 mod q, or mod r for polynomial coefficients), witness values are written as
 * they are.  Polynomial records follow JS key order (integer keys ascending), like the reference's loop
 * over Object.keys.
 */
"use strict";
const Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const MASK = (1n << 256n) - 1n;

function toBig(v) {
    if (typeof v === "bigint") return v;
    if (typeof v === "number" && Number.isSafeInteger(v) && v >= 0) return BigInt(v);
    if (typeof v === "string" && /^[0-9]+$/.test(v)) return BigInt(v);
    throw new TypeError("not a decimal integer: " + String(v));
}
function putLE32(buf, off, v) {
    v &= MASK;
    for (let i = 0; i < 4; i++) { buf.writeBigUInt64LE(v & 0xFFFFFFFFFFFFFFFFn, off + 8 * i); v >>= 64n; }
    return off + 32;
}
const montQ = (v) => (toBig(v) << 256n) % Q;
const montR = (v) => (toBig(v) << 256n) % R;

function polKeys(p) {
    if (p === null || typeof p !== "object" || Array.isArray(p)) throw new TypeError("polynomial must be an object {constraint: coefficient}");
    const keys = Object.keys(p);
    for (const k of keys) if (!/^[0-9]+$/.test(k)) throw new TypeError("polynomial key is not a constraint index: " + k);
    return keys;
}

function pkeyJsonToBin(pk) {
    const nVars = Number(toBig(pk.nVars)), nPublic = Number(toBig(pk.nPublic)), domain = Number(toBig(pk.domainSize));
    for (const name of ["polsA", "polsB", "A", "B1", "B2", "C"])
        if (!Array.isArray(pk[name]) || pk[name].length < nVars) throw new RangeError(name + " has fewer than nVars entries");
    if (!Array.isArray(pk.hExps) || pk.hExps.length < domain) throw new RangeError("hExps has fewer than domainSize entries");
    let size = 40 + 3 * 64 + 2 * 128;
    for (let i = 0; i < nVars; i++) size += 8 + 36 * (polKeys(pk.polsA[i]).length + polKeys(pk.polsB[i]).length);
    size += nVars * (64 + 64 + 128) + (nVars - nPublic - 1) * 64 + domain * 64;
    if (size > 0xFFFFFFFF) throw new RangeError("key needs " + size + " bytes: proving_key.bin uses u32 offsets (load it by sections instead)");
    const buf = Buffer.alloc(size);
    let o = 40;
    const g1 = (p) => { o = putLE32(buf, o, montQ(p[0])); o = putLE32(buf, o, montQ(p[1])); };
    const g2 = (p) => { o = putLE32(buf, o, montQ(p[0][0])); o = putLE32(buf, o, montQ(p[0][1])); o = putLE32(buf, o, montQ(p[1][0])); o = putLE32(buf, o, montQ(p[1][1])); };
    const pol = (p) => {
        const keys = polKeys(p);
        buf.writeUInt32LE(keys.length, o); o += 4;
        for (const k of keys) { buf.writeUInt32LE(Number(BigInt(k) & 0xFFFFFFFFn), o); o += 4; o = putLE32(buf, o, montR(p[k])); }
    };
    buf.writeUInt32LE(nVars, 0); buf.writeUInt32LE(nPublic, 4); buf.writeUInt32LE(domain, 8);
    g1(pk.vk_alfa_1); g1(pk.vk_beta_1); g1(pk.vk_delta_1); g2(pk.vk_beta_2); g2(pk.vk_delta_2);
    buf.writeUInt32LE(o, 12); for (let i = 0; i < nVars; i++) pol(pk.polsA[i]);
    buf.writeUInt32LE(o, 16); for (let i = 0; i < nVars; i++) pol(pk.polsB[i]);
    buf.writeUInt32LE(o, 20); for (let i = 0; i < nVars; i++) g1(pk.A[i]);
    buf.writeUInt32LE(o, 24); for (let i = 0; i < nVars; i++) g1(pk.B1[i]);
    buf.writeUInt32LE(o, 28); for (let i = 0; i < nVars; i++) g2(pk.B2[i]);
    buf.writeUInt32LE(o, 32); for (let i = nPublic + 1; i < nVars; i++) g1(pk.C[i]);
    buf.writeUInt32LE(o, 36); for (let i = 0; i < domain; i++) g1(pk.hExps[i]);
    if (o !== size) throw new Error("internal: wrote " + o + " of " + size + " bytes");
    return buf;
}

function witnessJsonToBin(w) {
    if (!Array.isArray(w)) throw new TypeError("witness must be an array");
    const buf = Buffer.alloc(32 * w.length);
    let o = 0;
    for (const v of w) o = putLE32(buf, o, toBig(v));
    return buf;
}

module.exports = { pkeyJsonToBin, witnessJsonToBin };
