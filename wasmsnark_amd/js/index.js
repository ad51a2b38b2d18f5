/*
 * Drop-in for wasmsnark's BN128 prover API on an MI355X (see ../../INTEGRATION.md).
 *
 * Same names and shapes as the reference:
 *   buildBn128() -> Promise<Bn128>                          (reference index.js:21, src/bn128.js:173-265)
 *   buildBn128({devices: [0, 1, ...]}) -> the same object over SEVERAL GPUs of this one process: the reference's worker pool
 *       (src/bn128.js:173-265: `build()` starts its workers; :353-415, :607-622: every sum is cut into one contiguous range of the
 *       pairs per worker and the partial results are added) with GPUs as the workers -- a points shard of the key per device,
 *       CALC_H on the distributed transform, the transport inside libwsnark.so (csrc/group.hip)
 *   Bn128.groth16GenProof(signals, pkey) -> Promise<{pi_a, pi_b, pi_c}> of decimal strings (src/bn128.js:580-720)
 *   Bn128.g1_multiexp / g2_multiexp / calcH / terminate      (src/bn128.js:353-415, 569-578, 562-566)
 *   groth16GenProof(witness, provingKey[, cb]) and the README name genZKSnarkProof (main_bn128.js:26-39, README.md:28-30)
 * Inputs may be ArrayBuffer (what the reference requires), Buffer or TypedArray.
 * All arithmetic runs in libwsnark.so (hand-written HIP); this file only marshals and formats.
 */
"use strict";
const addon = require("./build/wsnark_napi.node");
addon.g1g2 = (which, scalars, points) => (which ? addon.g2Multiexp(scalars, points) : addon.g1Multiexp(scalars, points));

// bin2int / bin2g1 / bin2g2 (src/bn128.js:319-351, 714-718): the twelve 256-bit integers of a proof as decimal strings, formatted in
// the addon (native code, no BigInt: ~5 us)
function proofFromBytes(ab) { return addon.proofToObject(ab); }

/* Two checks that a cached key handle still describes the bytes the caller is holding.
 *   fingerprint(u8): synchronous, a few KiB -- the 488-byte header + fixed points, 64 samples of 32 bytes spread over the buffer and
 *                    its tail.  It catches a buffer that was refilled with another key.
 *   digest(u8):      ALL of the bytes (addon.hashBytes, off the event loop, several threads; ~10 ms for a 0.6 GB key).
 * The reference re-parses pkey inside every call (src/bn128.js:581-604): a caller may patch a few bytes of a key between two proofs
 * and is served with the new key.  The DEFAULT here gives the same guarantee (round 5): every call that is handed key BYTES takes
 * the digest of all of them -- groth16GenProof takes it BESIDE the proof on the cached handle and, should it differ from the
 * digest taken when that handle was loaded, loads the new bytes and proves again (the stale proof is never returned).  Callers who
 * know their key bytes do not change opt into the sampled check alone with {trustCache: true}; callers who want no check at all hold
 * the handle of loadKey() and pass that. */
async function digest(u8) {
    return Buffer.from(await addon.hashBytes(u8)).toString("hex");
}
function fingerprint(u8) {
    const n = u8.byteLength;
    const b = Buffer.from(u8.buffer, u8.byteOffset, n);          // (a view: no copy)
    let h1 = 0x811c9dc5 | 0, h2 = 0x9e3779b9 | 0;                 // two 32-bit multiply-xor lanes over the sampled words (no BigInt: microseconds)
    const take = (off, len) => {
        const end = Math.min(off + len, n);
        let i = off;
        for (; i + 4 <= end; i += 4) {
            const w = b.readUInt32LE(i);
            h1 = Math.imul(h1 ^ w, 0x01000193);
            h2 = Math.imul((h2 ^ w) + ((h2 << 13) | (h2 >>> 19)), 0x85ebca6b);
        }
        for (; i < end; i++) { h1 = Math.imul(h1 ^ b[i], 0x01000193); h2 = Math.imul(h2 ^ b[i], 0x85ebca6b); }
    };
    take(0, 488);
    const step = Math.max(32, Math.floor(n / 64 / 32) * 32);
    for (let k = 1; k <= 64; k++) if (k * step <= n) take(k * step - 32, 32);
    take(Math.max(0, n - 64), 64);
    return (h1 >>> 0).toString(16) + (h2 >>> 0).toString(16) + ":" + n;
}
function asBytes(x) {
    if (x instanceof ArrayBuffer) return new Uint8Array(x);
    if (ArrayBuffer.isView(x)) return new Uint8Array(x.buffer, x.byteOffset, x.byteLength);
    throw new TypeError("expected an ArrayBuffer, Buffer or TypedArray");
}

class Bn128 {
    constructor(deviceInfo, group, devices) {
        this.deviceInfo = deviceInfo;
        this._group = group || null;      // several GPUs: wsnark_group_* (keys are points shards over the devices, proofs run on all of them)
        this.devices = devices || null;
        // proving-key OBJECT (the exact ArrayBuffer / view the caller passed) -> {handle (a Promise), byteOffset, byteLength,
        // fp (sampled fingerprint), digest (Promise of the whole-buffer digest taken at load time)}.
        // Two views of one ArrayBuffer (sub-arrays of a bundle, Node's pooled small Buffers) are different keys here.
        // A cached handle is reused while offset, length and the sampled fingerprint still match -- and, for callers that
        // ask ({trustCache: false}), the digest of the WHOLE buffer; see fingerprint() / digest() above.
        this._keys = new WeakMap();
        this._files = new Map();    // key FILE path -> {size, mtimeMs, handle (a Promise)}
        this.fullDigests = 0;   // how many whole-buffer digests this object has computed (tests, tools/node_bench.js)
        this._pr = null;     // blinding values of the last proof, "for tests" like the reference (src/bn128.js:662-664)
        this._ps = null;
        this._live = true;
    }
    /* points: the bytes (as in the reference, src/bn128.js:353-415), or a handle from loadPoints() */
    g1_multiexp(scalars, points) { return this._msm(0, scalars, points); }
    g2_multiexp(scalars, points) { return this._msm(1, scalars, points); }
    _msm(which, scalars, points) {
        if (!this._live) return Promise.reject(new Error("wsnark: this Bn128 object has been terminated"));
        if (points !== null && typeof points === "object" && !(points instanceof ArrayBuffer) && !ArrayBuffer.isView(points)) return addon.pointsMultiexp(points, scalars);
        return this._group ? addon.groupMultiexp(this._group, which, scalars, points) : addon.g1g2(which, scalars, points);
    }
    /* No counterpart in the reference: bases that are summed over again and again (a prover's key sections) made RESIDENT once, as
     * fixed-base window tables -- later g1_multiexp / g2_multiexp calls that pass the handle instead of the bytes pay neither the
     * points' upload nor the per-window plans (about half the time of a 2^20 sum).  group: 1 or 2. */
    loadPoints(group, points) { return addon.loadPoints(group, points); }
    calcH(signals, polsA, polsB, nSignals, domainSize) { return addon.calcH(signals, polsA, polsB, nSignals, domainSize); }
    fft(buf, odd) { return addon.fft(buf, odd | 0, false); }
    ifft(buf, odd) { return addon.fft(buf, odd | 0, true); }
    /* the cache entry of a key object if offset, length and the sampled fingerprint still match its bytes */
    _cached(pkey, u8) {
        const hit = this._keys.get(pkey);
        return hit && hit.byteOffset === u8.byteOffset && hit.byteLength === u8.byteLength && hit.fp === fingerprint(u8) ? hit : null;
    }
    /* pkey bytes -> device-resident key handle (stays in HBM across proofs; freed by the GC).
     * A cached handle is returned when the digest of ALL bytes equals the one taken when it was loaded; opts.trustCache === true: when
     * the sampled fingerprint does.  Concurrent callers with the same key object share ONE load: the entry is in the map before it is awaited. */
    async loadKey(pkey, opts) {
        if (!this._live) throw new Error("wsnark: this Bn128 object has been terminated");
        if (typeof pkey === "string") return this._loadKeyFile(pkey);
        if (pkey !== null && typeof pkey === "object" && !(pkey instanceof ArrayBuffer) && !ArrayBuffer.isView(pkey)) return pkey;   // already a handle
        const u8 = asBytes(pkey);
        const hit = this._cached(pkey, u8);
        if (hit) {
            if (opts && opts.trustCache === true) return hit.handle;                     // (a Promise: resolved, or the load in flight)
            this.fullDigests++;
            if ((await hit.digest) === (await digest(u8))) return hit.handle;
        }
        // The digest is taken beside the load (addon.hashBytes deals the buffer's blocks to several threads: ~10 ms for a 0.6 GB
        // key, a quarter of the load) and loadKey() resolves only when BOTH are done: a caller who rewrites bytes in place as
        // soon as the first call returns must find them compared against the bytes that were loaded, not against a digest that
        // was still being taken while they changed.
        const load = this._group ? addon.groupLoadKey(this._group, pkey) : addon.loadKey(pkey);
        const entry = { byteOffset: u8.byteOffset, byteLength: u8.byteLength, fp: fingerprint(u8), handle: load, digest: digest(u8) };
        this.fullDigests++;
        this._keys.set(pkey, entry);
        entry.digest.catch(() => {});
        try {
            const h = await entry.handle;
            await entry.digest;
            return h;
        } catch (e) {
            if (this._keys.get(pkey) === entry) this._keys.delete(pkey);                // a key that failed to parse is not cached
            throw e;
        }
    }
    /* A key FILE -- the reference's proving_key.bin or the WSNARK64 container for keys beyond its 4 GiB (js/formats.js:
     * writeKeyContainer; 2^24 constraints = 7.8 GB, more than one Buffer holds).  The library maps the file and reads only what it makes
     * resident; with a group every device reads its own shard.  Handles are cached per path while the file's size and mtime stand. */
    async _loadKeyFile(path) {
        const fs = require("fs");
        const st = fs.statSync(path);
        const hit = this._files.get(path);
        if (hit && hit.size === st.size && hit.mtimeMs === st.mtimeMs) return hit.handle;
        const entry = { size: st.size, mtimeMs: st.mtimeMs, handle: this._group ? addon.groupLoadKeyFile(this._group, path) : addon.loadKeyFile(path) };
        this._files.set(path, entry);
        try { return await entry.handle; } catch (e) { if (this._files.get(path) === entry) this._files.delete(path); throw e; }
    }
    /* {nVars, nPublic, domainSize, fileBytes, format} of a key file, from its header (nothing is loaded) */
    keyFileInfo(path) { return addon.keyFileInfo(path); }
    /* forget the cached handle of a key object (its bytes were rewritten in place, or its HBM should go back) */
    invalidateKey(pkey) { return typeof pkey === "string" ? this._files.delete(pkey) : this._keys.delete(pkey); }
    /* {nVars, nPublic, domainSize, loadMs: {polsToCsr, pointsH2d, masksConvert, tableBuild, total}} of a key (bytes or handle) */
    async keyInfo(pkey) { const h = await this.loadKey(pkey); return this._group ? addon.groupKeyInfo(h) : addon.keyInfo(h); }
    /* loadKey() returns once the key's sections are resident: proofs may start at once and run on the plain sections while the rows
     * of the fixed-base tables are built behind them (about 0.2 s for a 2^20 key).  A caller that wants its first timed proof at the
     * steady-state rate awaits this first. */
    async waitTables(pkey) { const h = await this.loadKey(pkey); return (this._group ? addon.groupWaitTables(h) : addon.waitTables(h)).then(() => true); }
    /* an ArrayBuffer of `bytes` bytes in PINNED host memory: a witness written into it is DMA'd to the GPU in place, chunk by
     * chunk, instead of being copied through the library's staging ring first (no counterpart in the reference) */
    allocInput(bytes) { return addon.allocPinned(bytes); }
    /* pkey: proving_key.bin bytes, or a handle from loadKey().
     * opts.r / opts.s: optional 32-byte blinding values (the reference draws them from crypto.randomBytes)
     * opts.trustCache: see loadKey.  opts.timing: an object that receives {loadKey_ms, prove_ms, format_ms} of this call */
    async groth16GenProof(signals, pkey, opts) {
        if (!this._live) throw new Error("wsnark: this Bn128 object has been terminated");
        const t0 = process.hrtime.bigint();
        // which entry point a handle goes to is fixed when the call STARTS: terminate() clears this._group while calls may still be
        // awaiting their key, and a group key must never reach the single-context prove()
        const proveFn = this._group ? addon.groupProve : addon.prove;
        const prove = (h) => proveFn(h, signals, opts && opts.r ? opts.r : null, opts && opts.s ? opts.s : null);
        const isBytes = pkey instanceof ArrayBuffer || ArrayBuffer.isView(pkey);      // (else: a handle, or the path of a key file)
        let t1, out;
        const hit = isBytes && !(opts && opts.trustCache === true) ? this._cached(pkey, asBytes(pkey)) : null;
        if (hit) {
            // key BYTES with a cached handle: prove on it at once, take the digest of all bytes beside the proof, and only return the
            // proof if those are the bytes the handle was loaded from (else: load what the caller holds now and prove again)
            this.fullDigests++;
            const h = await hit.handle;
            t1 = process.hrtime.bigint();
            const first = prove(h);
            first.catch(() => {});
            // The digest's eight threads read the key at the host's memory bandwidth; started together with the proof they compete with
            // the witness's staging copy (the first ~1 ms of the call) -- 0.15 ms per call on most boxes, 1.7 ms on one.  For keys large
            // enough for that to matter the digest starts a timer tick later: it still ends well before the proof does.
            const bytes = asBytes(pkey);
            const now = bytes.byteLength >= (64 << 20) ? new Promise((res) => setTimeout(res, 1)).then(() => digest(bytes)) : digest(bytes);
            now.catch(() => {});
            if ((await now) === (await hit.digest)) out = await first;
            else {
                await first.catch(() => {});
                if (this._keys.get(pkey) === hit) this._keys.delete(pkey);
                out = await prove(await this.loadKey(pkey, opts));
            }
        } else {
            const h = await this.loadKey(pkey, opts);
            t1 = process.hrtime.bigint();
            out = await prove(h);
        }
        const t2 = process.hrtime.bigint();
        this._pr = new Uint8Array(out.slice(384, 416));
        this._ps = new Uint8Array(out.slice(416, 448));
        const proof = proofFromBytes(out.slice(0, 384));
        if (opts && opts.timing) {
            const ms = (a, b) => Number(b - a) / 1e6;
            Object.assign(opts.timing, { loadKey_ms: ms(t0, t1), prove_ms: ms(t1, t2), format_ms: ms(t2, process.hrtime.bigint()) });
        }
        return proof;
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
        this._files.clear();
        if (this._group) { addon.groupFree(this._group); this._group = null; }     // (a group's contexts belong to this object alone)
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
/* buildBn128([device]) | buildBn128({devices: [...]}).  The addon binds the in-tree libwsnark.so and nothing else: there is no
 * library-path option and no CPU path in the product (the test-suite's emulator host is its own build of the addon under tests/). */
async function buildBn128(device, opts) {
    if (device !== null && typeof device === "object") { opts = device; device = undefined; }      // buildBn128({devices: [...]})
    const devices = opts && opts.devices ? Array.from(opts.devices, (d) => d | 0) : null;
    if (devices && devices.length === 0) throw new TypeError("devices: expected at least one device ordinal");
    // (the default context serves calcH / fft and the pinned input buffers; with a group it sits on the group's first device)
    const info = addon.init(devices ? devices[0] : (device === undefined || device === null ? -1 : device));
    liveInstances++;
    if (!devices) return new Bn128(info);
    try {
        return new Bn128(info + " x" + devices.length + " (devices " + devices.join(", ") + ")", addon.groupCreate(devices), devices);
    } catch (e) {
        if (--liveInstances === 0) addon.shutdown();
        throw e;
    }
}
/* The module-level calls share one Bn128 object that is never terminated on its own, exactly like the reference's
 * (main_bn128.js:26-39 builds its singleton and leaves the workers running; src/bn128.js:562-566 needs an explicit call): a
 * process that only uses these forms ends with terminate() below (the reference has no such export), or process.exit(). */
function terminate() {
    if (singleton) { singleton.terminate(); singleton = null; }
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
module.exports = { buildBn128, groth16GenProof, genZKSnarkProof: groth16GenProof, groth16Verify, terminate, Bn128, proofFromBytes,
    pkeyJsonToBin: formats.pkeyJsonToBin, witnessJsonToBin: formats.witnessJsonToBin,
    pkeyBinSections: formats.pkeyBinSections, writeKeyContainer: formats.writeKeyContainer, pkeyBinToContainer: formats.pkeyBinToContainer };
