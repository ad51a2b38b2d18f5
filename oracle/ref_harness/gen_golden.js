/*
 * gen_golden.js -- runs the READ-ONLY reference (prebuilt WASM + src/bn128.js,
 * loaded by path through refenv.js) on seeded inputs and writes golden
 * input/output vectors to tests/golden/*.json.  TEST INFRASTRUCTURE ONLY.
 *
 *   node oracle/ref_harness/gen_golden.js            # primitives
 *   node oracle/ref_harness/gen_golden.js proofs     # proofs for tests/golden/keys/*.bin
 *   node oracle/ref_harness/gen_golden.js unreduced  # CALC_H and proofs with witness values / key coefficients in [r, 2^256)
 *
 * The vectors are DATA (inputs + the reference's outputs); no reference source
 * is copied.  Runs only in the build container (the reference is absent on the
 * GPU box); the committed JSON files are what travels.
 */
"use strict";
const fs = require("fs");
const path = require("path");
const E = require("./refenv.js");
const { hex, unhex, le32, fromLE, Rng, toAB } = E;

const OUT = path.join(__dirname, "..", "..", "tests", "golden");
const Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;

function edgeSet(p) {
    // test/f1.js:296-311 edge set + extras
    const h = (p - 1n) / 2n;
    return [0n, 1n, 2n, p - 1n, p - 2n, h, h + 1n, h + 2n, h - 1n, h - 2n, (1n << 64n) - 1n, 1n << 64n,
        (1n << 128n) + 1n, (1n << 192n) - 1n, (1n << 253n), p - (1n << 32n)];
}

async function primitives() {
    const { bn, ex, statics } = await E.buildRef();
    const H = E.mkHelpers(bn);
    const rng = new Rng(20240926);
    const pa = H.alloc(192), pb = H.alloc(192), pc = H.alloc(192);

    /* ---------------- fields ---------------- */
    const fields = {};
    for (const [name, pfx, p] of [["fq", "f1m", Q], ["fr", "frm", R]]) {
        const vals = edgeSet(p);
        for (let i = 0; i < 24; i++) vals.push(rng.below(p));
        const un = [], bin = [];
        for (const a of vals) {
            H.put(pa, le32(a));
            const o = { a: hex(le32(a)) };
            for (const op of ["square", "neg", "toMontgomery", "fromMontgomery", "inverse"]) {
                if (op === "inverse" && a === 0n) continue;
                ex[pfx + "_" + op](pa, pc);
                o[op] = hex(H.get(pc, 32));
            }
            un.push(o);
        }
        for (let i = 0; i < vals.length; i++) {
            for (const j of [i, (i * 7 + 3) % vals.length, (i * 13 + 5) % vals.length]) {
                const a = vals[i], b = vals[j];
                H.put(pa, le32(a)); H.put(pb, le32(b));
                const o = { a: hex(le32(a)), b: hex(le32(b)) };
                for (const op of ["mul", "add", "sub"]) { ex[pfx + "_" + op](pa, pb, pc); o[op] = hex(H.get(pc, 32)); }
                bin.push(o);
            }
        }
        fields[name] = { unary: un, binary: bin };
    }
    // test/f1.js:355-372: toMontgomery(11) over r
    /* ---------------- Fq2 ---------------- */
    const f2 = [];
    for (let i = 0; i < 24; i++) {
        const a = i < 4 ? [[0n, 1n], [1n, 0n], [Q - 1n, Q - 1n], [0n, 0n]][i] : [rng.below(Q), rng.below(Q)];
        const b = [rng.below(Q), rng.below(Q)];
        H.put(pa, le32(a[0])); H.put(pa + 32, le32(a[1])); H.put(pb, le32(b[0])); H.put(pb + 32, le32(b[1]));
        const o = { a: hex(H.get(pa, 64)), b: hex(H.get(pb, 64)) };
        ex.f2m_mul(pa, pb, pc); o.mul = hex(H.get(pc, 64));
        ex.f2m_square(pa, pc); o.square = hex(H.get(pc, 64));
        if (!(a[0] === 0n && a[1] === 0n)) { ex.f2m_inverse(pa, pc); o.inverse = hex(H.get(pc, 64)); }
        f2.push(o);
    }

    /* ---------------- groups ---------------- */
    const groups = {};
    for (const [name, pfx, sz, pgen] of [["g1", "g1m", 96, statics.pG1gen], ["g2", "g2m", 192, statics.pG2gen]]) {
        const P = H.alloc(sz), Qp = H.alloc(sz), T = H.alloc(sz), Z = H.alloc(sz), sc = H.alloc(64);
        const mk = (k, dst) => { H.put(sc, le32(k)); ex[pfx + "_timesScalar"](pgen, sc, 32, dst); };
        const cases = [];
        const dump = (p) => hex(H.get(p, sz));
        const aff = (p) => { ex[pfx + "_affine"](p, T); return hex(H.get(T, sz)); };
        const addCase = (label) => {
            const o = { label, p: dump(P), q: dump(Qp) };
            ex[pfx + "_add"](P, Qp, Z); o.add = dump(Z); o.add_affine = aff(Z); o.add_is_zero = ex[pfx + "_isZero"](Z);
            ex[pfx + "_double"](P, Z); o.double = dump(Z); o.double_affine = aff(Z);
            ex[pfx + "_neg"](P, Z); o.neg = dump(Z);
            o.p_affine = aff(P);
            o.eq = ex[pfx + "_eq"](P, Qp);
            cases.push(o);
        };
        mk(5n, P); mk(9n, Qp); addCase("generic jacobian (non-unit z)");
        mk(rng.below(R), P); mk(rng.below(R), Qp); addCase("generic random");
        mk(7n, P); ex[pfx + "_copy"](P, Qp); addCase("P+P -> double branch");
        mk(7n, P); ex[pfx + "_double"](P, Qp); ex[pfx + "_add"](Qp, P, Qp); mk(21n, P); addCase("same point, different z");
        mk(11n, P); ex[pfx + "_neg"](P, Qp); addCase("P+(-P) -> z=0 fallthrough");
        ex[pfx + "_zero"](P); mk(3n, Qp); addCase("inf+Q");
        mk(3n, P); ex[pfx + "_zero"](Qp); addCase("P+inf");
        ex[pfx + "_zero"](P); ex[pfx + "_zero"](Qp); addCase("inf+inf");
        ex[pfx + "_copy"](pgen, P); mk(2n, Qp); addCase("G + 2G (z1 = 1)");
        // timesScalar vectors incl. scalar >= r and 64-byte scalars
        const ts = [];
        for (const k of [0n, 1n, 2n, 10n, R - 1n, R, R + 5n, (1n << 256n) - 1n, rng.big(256)]) {
            H.put(sc, le32(k)); ex[pfx + "_timesScalar"](pgen, sc, 32, Z);
            ts.push({ scalar: hex(le32(k)), bytes: 32, affine: aff(Z) });
        }
        {
            const k = rng.big(512); const b = new Uint8Array(64); let x = k; for (let i = 0; i < 64; i++) { b[i] = Number(x & 0xFFn); x >>= 8n; }
            H.put(sc, b); ex[pfx + "_timesScalar"](pgen, sc, 64, Z);
            ts.push({ scalar: hex(b), bytes: 64, affine: aff(Z) });
        }
        groups[name] = { gen: hex(H.get(pgen, sz)), cases, times_scalar: ts };
    }

    /* ---------------- MSM ---------------- */
    // points: affine Montgomery k_i*G (k_i random), with planted edge cases.
    const msm = { g1: [], g2: [] };
    for (const [name, pfx, sz, affsz, pgen] of [["g1", "g1m", 96, 64, statics.pG1gen], ["g2", "g2m", 192, 128, statics.pG2gen]]) {
        const T = H.alloc(sz), sc = H.alloc(32), res = H.alloc(sz);
        const mkAff = (k) => { H.put(sc, le32(k)); ex[pfx + "_timesScalar"](pgen, sc, 32, T); ex[pfx + "_affine"](T, T); return H.get(T, affsz); };
        const sizes = name === "g1" ? [0, 1, 6, 7, 8, 15, 64, 300] : [0, 1, 6, 7, 8, 15, 64];
        for (const n of sizes) {
            for (const flavour of ["uniform", "edge"]) {
                if (n === 0 && flavour === "edge") continue;
                const scal = new Uint8Array(n * 32), pts = new Uint8Array(n * affsz);
                for (let i = 0; i < n; i++) {
                    let s = rng.below(R), pt = mkAff(rng.below(R));
                    if (flavour === "edge") {
                        const m = i % 12;
                        if (m === 0) s = 0n;
                        else if (m === 1) s = 1n;
                        else if (m === 2) s = R - 1n;
                        else if (m === 3) s = (1n << 256n) - 1n - BigInt(i);    // >= r, raw 256-bit
                        else if (m === 4) s = rng.big(32);                        // small
                        else if (m === 5) { pt = new Uint8Array(affsz); }         // x == 0 -> infinity (all zero)
                        else if (m === 6) { pt = pt.slice(); pt.fill(0, 0, affsz / 2); } // x == 0, y != 0 -> infinity
                        else if (m === 7 && i >= 1) { pt = pts.slice((i - 1) * affsz, i * affsz); } // duplicate of previous point
                        else if (m === 8 && i >= 1) { // negation of previous point, same scalar -> cancels
                            H.put(T, pts.slice((i - 1) * affsz, i * affsz)); ex[(name === "g1" ? "f1m" : "f2m") + "_one"](T + affsz);
                            ex[pfx + "_neg"](T, T); pt = H.get(T, affsz); s = fromLE(scal.slice((i - 1) * 32, i * 32));
                        } else if (m === 9) s = R + BigInt(i);
                    }
                    scal.set(le32(s), i * 32); pts.set(pt, i * affsz);
                }
                const ps = H.putNew(scal), pp = H.putNew(pts);
                const o = { n, flavour, scalars: Buffer.from(scal).toString("base64"), points: Buffer.from(pts).toString("base64") };
                if (name === "g1") {
                    ex.g1m_zero(res); ex.g1m_multiexp2(ps, pp, n, 7, res); ex.g1m_affine(res, T); o.multiexp2_affine = hex(H.get(T, sz));
                    ex.g1m_zero(res); ex.g1m_multiexp(ps, pp, n, 7, res); ex.g1m_affine(res, T); o.multiexp_affine = hex(H.get(T, sz));
                    if (n >= 8) { // host-level sharded path, src/bn128.js:353-383
                        const r96 = await bn.g1_multiexp(toAB(scal), toAB(pts));
                        o.host_affine = hex(new Uint8Array(bn.g1_affine(r96)));
                    }
                } else {
                    ex.g2m_zero(res); ex.g2m_multiexp(ps, pp, n, 7, res); ex.g2m_affine(res, T); o.multiexp_affine = hex(H.get(T, sz));
                    if (n >= 8) {
                        const r192 = await bn.g2_multiexp(toAB(scal), toAB(pts));
                        o.host_affine = hex(new Uint8Array(bn.g2_affine(r192)));
                    }
                }
                msm[name].push(o);
            }
        }
        // accumulate-into-pr semantics: pr preset to 3G, n = 2
        {
            const scal = new Uint8Array(64), pts = new Uint8Array(2 * affsz);
            scal.set(le32(5n), 0); scal.set(le32(6n), 32); pts.set(mkAff(2n), 0); pts.set(mkAff(4n), affsz);
            H.put(sc, le32(3n)); ex[pfx + "_timesScalar"](pgen, sc, 32, res);
            const ps = H.putNew(scal), pp = H.putNew(pts);
            ex[name === "g1" ? "g1m_multiexp2" : "g2m_multiexp"](ps, pp, 2, 7, res); ex[pfx + "_affine"](res, T);
            msm[name].push({ n: 2, flavour: "accumulate_into_3G", scalars: Buffer.from(scal).toString("base64"),
                points: Buffer.from(pts).toString("base64"), acc_affine: hex(H.get(T, sz)) });
        }
    }

    /* ---------------- FFT ---------------- */
    const fft = [];
    for (const n of [1, 2, 4, 8, 64, 1024]) {
        const x = new Uint8Array(n * 32);
        for (let i = 0; i < n; i++) x.set(le32(n <= 4 ? BigInt(i) : rng.below(R)), i * 32);
        const p = H.alloc(n * 32), pm = H.alloc(n * 32);
        H.put(p, x); ex.fft_toMontgomeryN(p, pm, n);
        const xm = H.get(pm, n * 32);
        const o = { n, input_mont: Buffer.from(xm).toString("base64") };
        for (const [label, fn, odd] of [["fft0", "fft_fft", 0], ["fft1", "fft_fft", 1], ["ifft0", "fft_ifft", 0], ["ifft1", "fft_ifft", 1]]) {
            // fft_ifft(n=1) never terminates cleanly in the reference (__finalInverse loops from i=1
            // until i == n/2 == 0, build_fft.js:575-583, and runs out of memory bounds): recorded as a trap
            if (n === 1 && fn === "fft_ifft") { o[label] = null; continue; }
            H.put(pm, xm); ex[fn](pm, n, odd); o[label] = Buffer.from(H.get(pm, n * 32)).toString("base64");
        }
        if (n === 4) { H.put(pm, xm); ex.fft_fft(pm, n, 0); ex.fft_fromMontgomeryN(pm, p, n); o.fft0_plain = hex(H.get(p, n * 32)); }
        fft.push(o);
    }
    const fftTraps = [];
    for (const n of [0, 3, 6, 1000]) {
        let threw = false; const p = H.alloc(Math.max(n, 1) * 32);
        try { ex.fft_fft(p, n, 0); } catch (e) { threw = true; }
        fftTraps.push({ n, traps: threw });
    }

    /* ---------------- CALC_H (worker command) ---------------- */
    const calch = [];
    for (const [nSignals, domain, maxnnz] of [[5, 4, 2], [20, 16, 3], [100, 64, 3], [37, 64, 1]]) {
        const sig = new Uint8Array(nSignals * 32);
        for (let i = 0; i < nSignals; i++) sig.set(le32(i === 0 ? 1n : (i % 5 === 0 ? rng.big(32) : rng.below(R))), i * 32);
        const mkPols = () => {
            const parts = [];
            for (let s = 0; s < nSignals; s++) {
                const k = Number(rng.next64() % BigInt(maxnnz + 1));
                const hdr = new Uint8Array(4); new DataView(hdr.buffer).setUint32(0, k, true); parts.push(hdr);
                const used = new Set();
                for (let j = 0; j < k; j++) {
                    let idx; do { idx = Number(rng.next64() % BigInt(domain)); } while (used.has(idx)); used.add(idx);
                    const rec = new Uint8Array(36); new DataView(rec.buffer).setUint32(0, idx, true);
                    rec.set(le32((rng.below(R) << 256n) % R), 4);   // Montgomery form of a random coef
                    parts.push(rec);
                }
            }
            return Buffer.concat(parts.map((p) => Buffer.from(p)));
        };
        const pA = mkPols(), pB = mkPols();
        const h = await bn.calcH(toAB(sig), toAB(pA), toAB(pB), nSignals, domain);
        calch.push({ nSignals, domain, signals: Buffer.from(sig).toString("base64"), polsA: pA.toString("base64"),
            polsB: pB.toString("base64"), h: Buffer.from(new Uint8Array(h)).toString("base64") });
    }

    fs.mkdirSync(OUT, { recursive: true });
    const W = (f, o) => fs.writeFileSync(path.join(OUT, f), JSON.stringify(o, null, 0));
    W("fields.json", { q: Q.toString(), r: R.toString(), fields, fq2: f2 });
    W("groups.json", groups);
    W("msm.json", msm);
    W("fft.json", { cases: fft, traps: fftTraps });
    W("calch.json", calch);
    bn.terminate();
}

/* proofs for every tests/golden/keys/<name>.{pkey.bin,witness.bin,vk.json,public.json} */
async function proofs() {
    const { bn } = await E.buildRef();
    const dir = path.join(OUT, "keys");
    const names = fs.readdirSync(dir).filter((f) => f.endsWith(".pkey.bin")).map((f) => f.slice(0, -9)).sort();
    const rng = new Rng(777);
    const out = {};
    for (const name of names) {
        const pkey = fs.readFileSync(path.join(dir, name + ".pkey.bin"));
        const wit = fs.readFileSync(path.join(dir, name + ".witness.bin"));
        const vk = JSON.parse(fs.readFileSync(path.join(dir, name + ".vk.json"), "utf8"));
        const pub = JSON.parse(fs.readFileSync(path.join(dir, name + ".public.json"), "utf8"));
        if (!vk.vk_alfabeta_12) vk.vk_alfabeta_12 = [[["0", "0"], ["0", "0"], ["0", "0"]], [["0", "0"], ["0", "0"], ["0", "0"]]];
        const cases = [];
        const rs = [[0n, 0n], [7n, 9n], [(1n << 256n) - 1n, (1n << 256n) - 2n], [rng.big(256), rng.big(256)]];
        for (const [r, s] of rs) {
            E.setRS(le32(r), le32(s));
            const proof = await bn.groth16GenProof(toAB(new Uint8Array(wit)), toAB(new Uint8Array(pkey)));
            // read back what the reference really used (src/bn128.js:662-664)
            const rUsed = hex(new Uint8Array(bn.getBin(bn._pr, 32))), sUsed = hex(new Uint8Array(bn.getBin(bn._ps, 32)));
            if (rUsed !== hex(le32(r)) || sUsed !== hex(le32(s))) throw new Error("r,s injection failed");
            const ok = await bn.groth16Verify(vk, pub, proof);
            const bad = pub.length ? await bn.groth16Verify(vk, [String((BigInt(pub[0]) + 1n) % R)].concat(pub.slice(1)), proof) : null;
            cases.push({ r: rUsed, s: sUsed, proof, reference_verifies: ok, reference_rejects_wrong_public: bad === null ? null : !bad });
            console.log(name, "r=", r.toString(16).slice(0, 8), "verify:", ok, "wrong-public rejected:", bad === null ? "n/a" : !bad);
        }
        out[name] = cases;
    }
    fs.writeFileSync(path.join(OUT, "proofs.json"), JSON.stringify(out, null, 0));
    bn.terminate();
}

/* Inputs the reference accepts but nothing canonicalises before the prover sees them: witness values and key coefficients in
 * [r, 2^256).  fft_toMontgomeryN (src/build_fft.js:418-458) reduces the signals, g1m_multiexp2 / g2m_multiexp take the raw
 * 256-bit scalars (src/build_multiexp.js:651-744, 498-580), pol_constructLC multiplies whatever the key holds
 * (src/build_pol.js:62-144).  CALC_H instances and whole proofs on the t6 key, both with such values, from the reference itself. */
async function unreduced() {
    const { bn } = await E.buildRef();
    const rng = new Rng(5151);
    const top = (1n << 256n) - 1n;
    const lift = (v) => {                       // the same residue, somewhere in [r, 2^256)
        const kmax = (top - v) / R;             // >= 5
        const k = 1n + rng.next64() % kmax;
        return v + k * R;
    };
    const calch = [];
    for (const [nSignals, domain, maxnnz] of [[20, 16, 3], [100, 64, 3]]) {
        const sig = new Uint8Array(nSignals * 32);
        for (let i = 0; i < nSignals; i++) {
            const v = i === 0 ? 1n : rng.below(R);
            sig.set(le32(i % 3 === 1 ? lift(v) : (i % 7 === 3 ? top - BigInt(i) : v)), i * 32);
        }
        const mkPols = () => {
            const parts = [];
            for (let s = 0; s < nSignals; s++) {
                const k = Number(rng.next64() % BigInt(maxnnz + 1));
                const hdr = new Uint8Array(4); new DataView(hdr.buffer).setUint32(0, k, true); parts.push(hdr);
                const used = new Set();
                for (let j = 0; j < k; j++) {
                    let idx; do { idx = Number(rng.next64() % BigInt(domain)); } while (used.has(idx)); used.add(idx);
                    const rec = new Uint8Array(36); new DataView(rec.buffer).setUint32(0, idx, true);
                    const c = (rng.below(R) << 256n) % R;
                    rec.set(le32((s + j) % 2 ? lift(c) : c), 4);   // every other coefficient NOT canonical
                    parts.push(rec);
                }
            }
            return Buffer.concat(parts.map((p) => Buffer.from(p)));
        };
        const pA = mkPols(), pB = mkPols();
        const h = await bn.calcH(toAB(sig), toAB(pA), toAB(pB), nSignals, domain);
        calch.push({ nSignals, domain, signals: Buffer.from(sig).toString("base64"), polsA: pA.toString("base64"),
            polsB: pB.toString("base64"), h: Buffer.from(new Uint8Array(h)).toString("base64") });
    }
    // whole proofs on the t6 key: (a) witness values lifted, (b) the key's coefficients lifted, (c) both
    const dir = path.join(OUT, "keys");
    const pkey = new Uint8Array(fs.readFileSync(path.join(dir, "t6.pkey.bin")));
    const wit = new Uint8Array(fs.readFileSync(path.join(dir, "t6.witness.bin")));
    const vk = JSON.parse(fs.readFileSync(path.join(dir, "t6.vk.json"), "utf8"));
    const pub = JSON.parse(fs.readFileSync(path.join(dir, "t6.public.json"), "utf8"));
    if (!vk.vk_alfabeta_12) vk.vk_alfabeta_12 = [[["0", "0"], ["0", "0"], ["0", "0"]], [["0", "0"], ["0", "0"], ["0", "0"]]];
    const wit2 = new Uint8Array(wit);
    for (let i = 1; i < wit2.length / 32; i++) if (i % 2) wit2.set(le32(lift(fromLE(wit.subarray(i * 32, i * 32 + 32)))), i * 32);
    const u32 = new DataView(pkey.buffer, pkey.byteOffset, 40);
    const nVars = u32.getUint32(0, true), pPolsA = u32.getUint32(12, true);
    const pkey2 = new Uint8Array(pkey);
    let off = pPolsA, lifted = 0;
    for (let m = 0; m < 2; m++) {               // polsA, then polsB right behind it (tools/buildpkey.js:124-186)
        for (let sgn = 0; sgn < nVars; sgn++) {
            const k = new DataView(pkey2.buffer, off, 4).getUint32(0, true); off += 4;
            for (let j = 0; j < k; j++) {
                if ((sgn + j) % 2 === 0) { pkey2.set(le32(lift(fromLE(pkey2.subarray(off + 4, off + 36)))), off + 4); lifted++; }
                off += 36;
            }
        }
    }
    const proofsOut = [];
    const rs = [[7n, 9n], [(1n << 256n) - 1n, (1n << 256n) - 2n]];
    for (const [label, w, k] of [["witness lifted", wit2, pkey], ["coefficients lifted", wit, pkey2], ["both lifted", wit2, pkey2]]) {
        for (const [r, s] of rs) {
            E.setRS(le32(r), le32(s));
            const proof = await bn.groth16GenProof(toAB(new Uint8Array(w)), toAB(new Uint8Array(k)));
            const ok = await bn.groth16Verify(vk, pub, proof);
            proofsOut.push({ label, r: hex(le32(r)), s: hex(le32(s)), proof, reference_verifies: ok });
            console.log(label, "verify:", ok);
        }
    }
    fs.writeFileSync(path.join(OUT, "unreduced.json"), JSON.stringify({ calch, key: "t6", coefficients_lifted: lifted,
        witness_lifted: Buffer.from(wit2).toString("base64"), pkey_lifted: Buffer.from(pkey2).toString("base64"), proofs: proofsOut }, null, 0));
    bn.terminate();
}

const mode = process.argv[2] || "primitives";
(mode === "proofs" ? proofs() : mode === "unreduced" ? unreduced() : primitives()).catch((e) => { console.error(e); process.exit(1); });
