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

/* ---- the u64-offset container for keys beyond proving_key.bin's 4 GiB (SURVEY.md section 8(f)1; layout: csrc/keyfile.hip) ----
 * pkeyBinSections(buf) -> {nVars, nPublic, domainSize, alfa1, ..., pointsH}: views into a proving_key.bin image (true section bounds)
 * writeKeyContainer(path, sections) -> file length: the same sections behind 64-bit offsets, written synchronously section by section
 *   (each of them a Buffer / TypedArray / ArrayBuffer of any size; a key too large for one Buffer is written from its own pieces)
 * pkeyBinToContainer(buf, path): proving_key.bin bytes -> the container.  loadKey(path) / wsnark_pkey_load_file read it. */
const fs = require("fs");
const HEADER = 608, ALIGN = 4096;
const ORDER = ["polsA", "polsB", "pointsA", "pointsB1", "pointsB2", "pointsC", "pointsH"];      // tools/buildpkey.js:166-186
function bytesOf(v) {
    if (Buffer.isBuffer(v)) return v;
    if (v instanceof ArrayBuffer) return Buffer.from(v);
    if (ArrayBuffer.isView(v)) return Buffer.from(v.buffer, v.byteOffset, v.byteLength);
    throw new TypeError("expected a Buffer, TypedArray or ArrayBuffer");
}
function pkeyBinSections(key) {
    const b = bytesOf(key);
    if (b.length < 488) throw new RangeError("proving key shorter than its fixed header");
    const h = []; for (let i = 0; i < 10; i++) h.push(b.readUInt32LE(4 * i));
    const [nVars, nPublic, domainSize, pPA, pPB, pA, pB1, pB2, pC, pH] = h;
    if (nPublic + 1 > nVars || !(488 <= pPA && pPA <= pPB && pPB <= pA) || pH + domainSize * 64 > b.length || pB2 + nVars * 128 > b.length)
        throw new RangeError("proving key: section offsets out of range");
    return { nVars, nPublic, domainSize, alfa1: b.slice(40, 104), beta1: b.slice(104, 168), delta1: b.slice(168, 232), beta2: b.slice(232, 360),
        delta2: b.slice(360, 488), polsA: b.slice(pPA, pPB), polsB: b.slice(pPB, pA), pointsA: b.slice(pA, pA + nVars * 64),
        pointsB1: b.slice(pB1, pB1 + nVars * 64), pointsB2: b.slice(pB2, pB2 + nVars * 128), pointsC: b.slice(pC, pC + (nVars - nPublic - 1) * 64),
        pointsH: b.slice(pH, pH + domainSize * 64) };
}
function writeKeyContainer(path, sec) {
    const nVars = sec.nVars, nPublic = sec.nPublic, dom = sec.domainSize;
    const want = { pointsA: nVars * 64, pointsB1: nVars * 64, pointsB2: nVars * 128, pointsC: (nVars - nPublic - 1) * 64, pointsH: dom * 64 };
    const views = {};
    for (const k of ORDER) {
        views[k] = bytesOf(sec[k]);
        if (k in want) {
            if (views[k].length < want[k]) throw new RangeError("key container: section " + k + " is shorter than its header-implied " + want[k] + " bytes");
            views[k] = views[k].slice(0, want[k]);
        }
    }
    let off = HEADER;
    const offs = {};
    for (const k of ORDER) { off = Math.ceil(off / ALIGN) * ALIGN; offs[k] = off; off += views[k].length; }
    const total = off;
    const hdr = Buffer.alloc(HEADER);
    hdr.write("WSNARK64", 0, "latin1");
    [1, HEADER, nVars, nPublic, dom, 0].forEach((v, i) => hdr.writeUInt32LE(v, 8 + 4 * i));
    [offs.polsA, views.polsA.length, offs.polsB, views.polsB.length, offs.pointsA, offs.pointsB1, offs.pointsB2, offs.pointsC, offs.pointsH, total]
        .forEach((v, i) => hdr.writeBigUInt64LE(BigInt(v), 32 + 8 * i));
    let o = 160;
    for (const [k, n] of [["alfa1", 64], ["beta1", 64], ["delta1", 64], ["beta2", 128], ["delta2", 128]]) {
        const v = bytesOf(sec[k]);
        if (v.length !== n) throw new RangeError("key container: " + k + " must be " + n + " bytes");
        v.copy(hdr, o); o += n;
    }
    const fd = fs.openSync(path, "w");
    try {
        fs.writeSync(fd, hdr, 0, HEADER, 0);
        for (const k of ORDER) {
            const v = views[k];
            for (let lo = 0; lo < v.length; lo += 64 << 20) {
                const n = Math.min(64 << 20, v.length - lo);
                let done = 0;
                while (done < n) done += fs.writeSync(fd, v, lo + done, n - done, offs[k] + lo + done);
            }
        }
        fs.ftruncateSync(fd, total);
    } finally { fs.closeSync(fd); }
    return total;
}
function pkeyBinToContainer(key, path) { return writeKeyContainer(path, pkeyBinSections(key)); }

module.exports = { pkeyJsonToBin, witnessJsonToBin, pkeyBinSections, writeKeyContainer, pkeyBinToContainer };
