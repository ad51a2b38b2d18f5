/*
 * refenv.js -- loads the READ-ONLY reference (iden3/wasmsnark, by path) so the
 * build's own scripts can run it as a black box.  TEST INFRASTRUCTURE ONLY.
 * Nothing from the reference is copied: it is `require`d from WSNARK_REF
 * (default /root/reference), which exists only in the build container.
 *
 * Hooks (all installed BEFORE the reference is required, none edits it):
 *  - big-integer -> ./node_modules/big-integer (BigInt-backed stand-in)
 *  - crypto.randomBytes -> deterministic stub when setRS(r, s) was called
 *    (the reference draws r then s: src/bn128.js:655-658)
 *  - Worker.prototype.postMessage -> raises `init` pages in INIT messages
 *    (the stock allocator cannot grow enough at 2^20: src/bn128.js:68-77,256)
 */
"use strict";
const path = require("path");
const Module = require("module");

const REF = process.env.WSNARK_REF || "/root/reference";
process.env.NODE_PATH = path.join(__dirname, "node_modules") + (process.env.NODE_PATH ? ":" + process.env.NODE_PATH : "");
Module._initPaths();

const crypto = require("crypto");
const realRandomBytes = crypto.randomBytes;
let rsQueue = [];
crypto.randomBytes = function (n) {
    if (rsQueue.length && n === 32) {
        const b = rsQueue.shift();
        // fresh, exactly-32-byte ArrayBuffer: the reference reads `.buffer`
        const ab = new ArrayBuffer(32);
        new Uint8Array(ab).set(b);
        return Buffer.from(ab);
    }
    return realRandomBytes.apply(crypto, arguments);
};
function setRS(r32, s32) { rsQueue = [Uint8Array.from(r32), Uint8Array.from(s32)]; }

const wt = require("worker_threads");
let initPages = 0;
const realPost = wt.Worker.prototype.postMessage;
wt.Worker.prototype.postMessage = function (msg, transfer) {
    if (initPages && msg && msg.command === "INIT") msg.init = initPages;
    return realPost.call(this, msg, transfer);
};
function setInitPages(p) { initPages = p; }

async function buildRef(concurrency) {
    if (concurrency) global.navigator = { hardwareConcurrency: concurrency };
    const build = require(path.join(REF, "src", "bn128.js"));
    const bn = await build();
    for (const w of bn.workers) w.on("error", (e) => { console.error("worker error", e); process.exit(3); });
    const statics = require(path.join(REF, "build", "bn128_wasm.js"));
    return { bn, ex: bn.instance.exports, statics };
}

/* ---- helpers over the main-thread instance ---- */
function mkHelpers(bn) {
    const H = {};
    H.alloc = (n) => bn.alloc(n);
    H.put = (p, u8) => { new Uint8Array(bn.memory.buffer).set(u8, p); };
    H.get = (p, n) => new Uint8Array(bn.memory.buffer.slice(p, p + n));
    H.putNew = (u8) => { const p = bn.alloc(u8.length || 4); H.put(p, u8); return p; };
    return H;
}

const hex = (u8) => Buffer.from(u8).toString("hex");
const unhex = (s) => Uint8Array.from(Buffer.from(s, "hex"));
function toAB(u8) { const ab = new ArrayBuffer(u8.length); new Uint8Array(ab).set(u8); return ab; }

/* splitmix64 -> deterministic bytes */
function Rng(seed) {
    let s = BigInt(seed) & 0xFFFFFFFFFFFFFFFFn;
    const M = 0xFFFFFFFFFFFFFFFFn;
    this.next64 = () => {
        s = (s + 0x9E3779B97F4A7C15n) & M;
        let z = s;
        z = ((z ^ (z >> 30n)) * 0xBF58476D1CE4E5B9n) & M;
        z = ((z ^ (z >> 27n)) * 0x94D049BB133111EBn) & M;
        return z ^ (z >> 31n);
    };
    this.big = (bits) => { let v = 0n; for (let i = 0; i < bits; i += 64) v = (v << 64n) | this.next64(); return v & ((1n << BigInt(bits)) - 1n); };
    this.below = (m) => this.big(320) % m;
}
function le32(v) { const o = new Uint8Array(32); let x = BigInt(v); for (let i = 0; i < 32; i++) { o[i] = Number(x & 0xFFn); x >>= 8n; } return o; }
function fromLE(u8) { let v = 0n; for (let i = u8.length - 1; i >= 0; i--) v = (v << 8n) | BigInt(u8[i]); return v; }

module.exports = { REF, buildRef, mkHelpers, setRS, setInitPages, hex, unhex, toAB, Rng, le32, fromLE };
