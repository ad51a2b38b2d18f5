// synth.hip -- synthetic Groth16 circuits, witnesses and key scalars from KNOWN toxic waste (host code only).
//
// NOT part of the reference's prove path and with no counterpart in it: the reference ships no proving key (SURVEY.md
// fact 9: test/data/proving_key.bin is missing), so parity tests and bench.py prove on synthetic keys.  The Python
// generator (wasmsnark_amd/synth.py: make_circuit / setup / build_sections) is exact but pure-Python big-integer
// loops: 22 s for a 2^20 circuit, 395 s for 2^24 -- too slow to put BASELINE config 5 under the GPU tests.  This is
// the same construction in C++ on the host field (field.h): a 2^24 circuit in seconds.
//
// Construction (formats: /root/reference tools/buildpkey.js:124-240, tools/buildwitness.js:36-69; key semantics of
// snarkjs "groth"):
//   circuit   nConstraints = domain - nPublic - 1 multiplication rows, row c defines variable 1 + nFree + c as
//             (sum A w)(sum B w); then the nPublic + 1 input-binding rows old snarkjs appends (A = w_i, B = C = 0).
//             style 0 "columns" (SURVEY.md section 8d C4): every variable occurs in 1-3 rows of A and of B, rows left
//             empty get one fill-in term; style 1 "rows": 1-2 terms per row among the earlier variables (~40 % of the
//             variables then never occur in A resp. B: their key points are infinity).
//             style 2 "boolean" (round 6; what circom's bit decompositions look like): groups of 14 FREE bit variables b_i with
//             their booleanity rows b_i (b_i - 1) = 0 (A = b_i, B = b_i - w_0, C = 0), one recomposition row
//             (sum 2^i b_i) * 1 = v and one product row (k v + 1) * p_prev = p per group: 87.5 % of the variables are 0 / 1 in the
//             witness, 6 % are 14-bit values, 6 % full-size field elements; w_0's B polynomial has an entry in every booleanity
//             row (one very long column), the v never occur in B (their B1 / B2 points are infinity).
//   setup     tau, alpha, beta, gamma, delta; a_s = A_s(tau), b_s, c_s through the Lagrange basis over w_n.
//   key       the discrete logarithm of every key point (the points themselves: wsnark_g{1,2}_mul_base_batch on the GPU).
//   expected  the discrete logarithms (a, b, c) of the proof for given r, s: a size-independent closed form.
#include <string.h>

#include <algorithm>
#include <new>
#include <thread>
#include <vector>

#include "../../include/wsnark.h"
#include "field.h"
#include "rt.h"

namespace wsnark {

namespace {

struct Rng {                                   // xoshiro256**, seeded through splitmix64
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        for (auto& v : s) {
            seed += 0x9e3779b97f4a7c15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            v = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }   // n > 0
    Fe fr_nonzero() {                           // uniform in [1, r), plain form
        for (;;) {
            Fe v{{next(), next(), next(), next() & 0x3fffffffffffffffull}};
            const Fe p = Fr::modulus();
            bool lt = false;
            for (int i = 3; i >= 0; i--) { if (v.l[i] != p.l[i]) { lt = v.l[i] < p.l[i]; break; } }
            if (lt && !Fr::is_zero(v)) return v;
        }
    }
};

template <class Fn>
void parallel_for(uint64_t n, Fn fn) {          // fn(lo, hi) on disjoint ranges
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 32) nt = 32;
    if (nt < 1 || n < (1u << 14)) { fn((uint64_t)0, n); return; }
    std::vector<std::thread> th;
    const uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto& t : th) t.join();
}

struct Sparse {                                // one R1CS matrix, column-major (CSC) with a row-major index on top
    std::vector<uint64_t> col_ptr;             // n_vars + 1
    std::vector<uint32_t> row;                 // nnz: constraint index
    std::vector<Fe> coef;                      // nnz: Montgomery
    std::vector<uint64_t> row_ptr;             // domain + 1
    std::vector<uint64_t> row_ent;             // nnz: index into row[] / coef[]; the column is found through ent_col
    std::vector<uint32_t> ent_col;             // nnz: column of entry k
    uint64_t absent = 0;                       // columns without any entry
};

struct Triplet { uint32_t col, row; Fe coef; };

void finish_sparse(std::vector<Triplet>& t, uint32_t n_vars, uint32_t domain, Sparse* M) {
    const uint64_t nnz = t.size();
    M->col_ptr.assign((size_t)n_vars + 1, 0);
    for (auto& e : t) M->col_ptr[e.col + 1]++;
    for (uint32_t s = 0; s < n_vars; s++) { if (!M->col_ptr[s + 1]) M->absent++; M->col_ptr[s + 1] += M->col_ptr[s]; }
    M->row.resize(nnz); M->coef.resize(nnz); M->ent_col.resize(nnz);
    std::vector<uint64_t> cur(M->col_ptr.begin(), M->col_ptr.end() - 1);
    for (auto& e : t) { const uint64_t k = cur[e.col]++; M->row[k] = e.row; M->coef[k] = e.coef; M->ent_col[k] = e.col; }
    std::vector<Triplet>().swap(t);
    M->row_ptr.assign((size_t)domain + 1, 0);
    for (uint64_t k = 0; k < nnz; k++) M->row_ptr[M->row[k] + 1]++;
    for (uint32_t c = 0; c < domain; c++) M->row_ptr[c + 1] += M->row_ptr[c];
    M->row_ent.resize(nnz);
    std::vector<uint64_t> rc(M->row_ptr.begin(), M->row_ptr.end() - 1);
    for (uint64_t k = 0; k < nnz; k++) M->row_ent[rc[M->row[k]]++] = k;
}

Fe fe_small(uint64_t v) { return Fe{{v, 0, 0, 0}}; }

}  // namespace

struct SynthCircuit {
    uint32_t log_domain = 0, n_public = 0, domain = 0, n_cons = 0, n_free = 0, n_vars = 0;
    Sparse A, B;
    std::vector<Fe> w;                          // witness, Montgomery
    Fe tau, alpha, beta, gamma, delta, z;       // toxic waste and Z(tau) = tau^n - 1, Montgomery
    std::vector<Fe> a, b, c;                    // a_s(tau), b_s(tau), c_s(tau), Montgomery
    std::vector<uint32_t> out_row;              // style 2: the row that DEFINES variable s (C holds 1 * s there), or NO_ROW
};
static const uint32_t NO_ROW = 0xffffffffu;
static const uint32_t BOOL_BITS = 14;           // style 2: bits per group

static Fe coef_of(Rng& g) { return Fr::to_mont((g.next() >> 63) ? g.fr_nonzero() : fe_small(1 + g.below(7))); }

static int synth_build(uint32_t log_domain, uint32_t n_public, uint64_t cseed, uint64_t sseed, int style, SynthCircuit* S) {
    if (log_domain < 2 || log_domain > 27 || style < 0 || style > 2) return WS_ERR_ARG;
    const uint32_t domain = 1u << log_domain;
    if ((uint64_t)n_public + 2 > domain) { set_last_error("synth: domain too small for nPublic"); return WS_ERR_SIZE; }
    S->log_domain = log_domain; S->n_public = n_public; S->domain = domain;
    const uint32_t n_cons = S->n_cons = domain - n_public - 1;
    const uint32_t n_free = S->n_free = n_public + 2;                     // public inputs + two private seeds
    // style 2: groups of BOOL_BITS bits + v + p over BOOL_BITS + 2 rows each; rows left over repeat a booleanity row
    const uint32_t n_groups = style == 2 ? n_cons / (BOOL_BITS + 2) : 0;
    if (style == 2 && n_groups == 0) { set_last_error("synth: domain too small for the boolean style"); return WS_ERR_SIZE; }
    const uint32_t n_vars = S->n_vars = style == 2 ? 1 + n_free + n_groups * (BOOL_BITS + 2) : 1 + n_free + n_cons;
    Rng g(cseed);
    S->w.assign(n_vars, Fr::zero());
    S->w[0] = Fr::one();
    for (uint32_t i = 1; i <= n_free; i++) S->w[i] = Fr::to_mont((i % 3) ? g.fr_nonzero() : fe_small(1 + g.below(0xffffffffull)));
    std::vector<Triplet> tA, tB;
    if (style == 2) {
        S->out_row.assign(n_vars, NO_ROW);
        const Fe one = Fr::one(), minus_one = Fr::neg(Fr::one());
        tA.reserve((size_t)n_cons * 2); tB.reserve((size_t)n_cons * 2);
        uint32_t row = 0, p_prev = n_free;                                // (the last private seed starts the product chain)
        for (uint32_t gi = 0; gi < n_groups; gi++) {
            const uint32_t b0 = 1 + n_free + gi * (BOOL_BITS + 2), v = b0 + BOOL_BITS, pv = v + 1;
            uint64_t bits = g.next();
            if (gi % 5 == 4) bits &= g.next() & g.next();                 // every fifth group sparse in ones: values of 0 dominate there
            for (uint32_t i = 0; i < BOOL_BITS; i++) {
                const uint32_t b = b0 + i;
                S->w[b] = ((bits >> i) & 1) ? one : Fr::zero();
                tA.push_back(Triplet{b, row, one});                       // b * (b - 1) = 0
                tB.push_back(Triplet{b, row, one});
                tB.push_back(Triplet{0, row, minus_one});
                row++;
            }
            for (uint32_t i = 0; i < BOOL_BITS; i++) tA.push_back(Triplet{b0 + i, row, Fr::to_mont(fe_small((uint64_t)1 << i))});
            tB.push_back(Triplet{0, row, one});                           // (sum 2^i b_i) * 1 = v
            S->out_row[v] = row++;
            tA.push_back(Triplet{v, row, coef_of(g)});                    // (k v + 1) * p_prev = p  (+ 1: a recomposed 0 must not end the chain)
            tA.push_back(Triplet{0, row, one});
            tB.push_back(Triplet{p_prev, row, one});
            S->out_row[pv] = row++;
            p_prev = pv;
        }
        while (row < n_cons) {                                            // left-over rows: one more booleanity row of some bit
            const uint32_t b = 1 + n_free + (uint32_t)g.below(n_groups) * (BOOL_BITS + 2) + (uint32_t)g.below(BOOL_BITS);
            tA.push_back(Triplet{b, row, one});
            tB.push_back(Triplet{b, row, one});
            tB.push_back(Triplet{0, row, minus_one});
            row++;
        }
    } else if (style == 0) {
        for (int m = 0; m < 2; m++) {
            std::vector<Triplet>& t = m ? tB : tA;
            t.reserve((size_t)n_vars * 5 / 2);
            std::vector<uint8_t> row_used((size_t)n_cons, 0);
            for (uint32_t s = 0; s < n_vars; s++) {
                const uint32_t c_min = s > n_free ? s - n_free : 0;      // first row whose output variable comes after s
                if (c_min >= n_cons) continue;
                const uint32_t k = 1 + (uint32_t)g.below(3);
                uint32_t got[3];
                uint32_t ng = 0;
                for (uint32_t j = 0; j < k; j++) {
                    const uint32_t c = c_min + (uint32_t)g.below(n_cons - c_min);
                    bool dup = false;
                    for (uint32_t q = 0; q < ng; q++) dup = dup || got[q] == c;
                    if (dup) continue;
                    got[ng++] = c;
                    t.push_back(Triplet{s, c, coef_of(g)});
                    row_used[c] = 1;
                }
            }
            for (uint32_t c = 0; c < n_cons; c++)
                if (!row_used[c]) t.push_back(Triplet{(uint32_t)g.below((uint64_t)1 + n_free + c), c, coef_of(g)});
        }
    } else {
        for (uint32_t c = 0; c < n_cons; c++) {
            const uint32_t out = 1 + n_free + c;
            for (int m = 0; m < 2; m++) {
                std::vector<Triplet>& t = m ? tB : tA;
                const uint32_t k = 1 + (uint32_t)(g.next() >> 63);
                uint32_t first = 0xffffffffu;
                for (uint32_t j = 0; j < k; j++) {
                    const uint32_t s = (uint32_t)g.below(out);
                    if (s == first) continue;
                    first = s;
                    t.push_back(Triplet{s, c, coef_of(g)});
                }
            }
        }
    }
    for (uint32_t i = 0; i <= n_public; i++) tA.push_back(Triplet{i, n_cons + i, Fr::one()});   // input-binding rows
    finish_sparse(tA, n_vars, domain, &S->A);
    finish_sparse(tB, n_vars, domain, &S->B);
    // the witness: every row's output variable comes after all of its inputs, so one sequential sweep
    auto row_dot = [&](const Sparse& M, uint32_t c) {
        Fe acc = Fr::zero();
        for (uint64_t e = M.row_ptr[c]; e < M.row_ptr[c + 1]; e++) {
            const uint64_t k = M.row_ent[e];
            acc = Fr::add(acc, Fr::mul(M.coef[k], S->w[M.ent_col[k]]));
        }
        return acc;
    };
    if (style == 2) {
        std::vector<uint32_t> var_of_row((size_t)n_cons, NO_ROW);
        for (uint32_t s = 0; s < n_vars; s++) if (S->out_row[s] != NO_ROW) var_of_row[S->out_row[s]] = s;
        for (uint32_t c = 0; c < n_cons; c++) {
            const Fe prod = Fr::mul(row_dot(S->A, c), row_dot(S->B, c));
            if (var_of_row[c] != NO_ROW) S->w[var_of_row[c]] = prod;
            else if (!Fr::is_zero(prod)) { set_last_error("synth: a booleanity row is not satisfied"); return WS_ERR_ARG; }
        }
    } else {
        for (uint32_t c = 0; c < n_cons; c++) S->w[1 + n_free + c] = Fr::mul(row_dot(S->A, c), row_dot(S->B, c));
    }
    // ---- setup ----
    Rng t(sseed);
    S->tau = Fr::to_mont(t.fr_nonzero()); S->alpha = Fr::to_mont(t.fr_nonzero()); S->beta = Fr::to_mont(t.fr_nonzero());
    S->gamma = Fr::to_mont(t.fr_nonzero()); S->delta = Fr::to_mont(t.fr_nonzero());
    // L_c(tau) = (tau^n - 1)/n * w^c / (tau - w^c): one batch inversion per thread range
    Fe wn = Fr::to_mont(Fe{{0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull}});   // 5^((r-1)/2^28), src/build_fft.js:29-47
    for (uint32_t i = 28; i > log_domain; i--) wn = Fr::sqr(wn);
    Fe tn = S->tau;
    for (uint32_t i = 0; i < log_domain; i++) tn = Fr::sqr(tn);
    S->z = Fr::sub(tn, Fr::one());
    if (Fr::is_zero(S->z)) { set_last_error("synth: tau is a domain point"); return WS_ERR_ARG; }
    const Fe zn = Fr::mul(S->z, Fr::inv(Fr::to_mont(fe_small(domain))));
    std::vector<Fe> L(domain);
    parallel_for(domain, [&](uint64_t lo, uint64_t hi) {
        std::vector<Fe> pw(hi - lo), pre(hi - lo);
        Fe x = Fr::pow_u64(wn, lo), run = Fr::one();
        for (uint64_t i = lo; i < hi; i++) {
            pw[i - lo] = x;
            pre[i - lo] = run;
            run = Fr::mul(run, Fr::sub(S->tau, x));
            x = Fr::mul(x, wn);
        }
        Fe inv = Fr::inv(run);
        for (uint64_t i = hi; i-- > lo;) {
            const Fe di = Fr::mul(inv, pre[i - lo]);
            inv = Fr::mul(inv, Fr::sub(S->tau, pw[i - lo]));
            L[i] = Fr::mul(Fr::mul(zn, pw[i - lo]), di);
        }
    });
    S->a.resize(n_vars); S->b.resize(n_vars); S->c.assign(n_vars, Fr::zero());
    parallel_for(n_vars, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t s = lo; s < hi; s++) {
            for (int m = 0; m < 2; m++) {
                const Sparse& M = m ? S->B : S->A;
                Fe acc = Fr::zero();
                for (uint64_t k = M.col_ptr[s]; k < M.col_ptr[s + 1]; k++) acc = Fr::add(acc, Fr::mul(M.coef[k], L[M.row[k]]));
                (m ? S->b : S->a)[s] = acc;
            }
            if (style == 2) { if (S->out_row[s] != NO_ROW) S->c[s] = L[S->out_row[s]]; }
            else if (s >= (uint64_t)1 + n_free) S->c[s] = L[s - 1 - n_free];      // C: row c holds 1 * (its output variable)
        }
    });
    return WS_OK;
}

static void store_plain_fr(uint8_t* dst, const Fe& mont) {
    const Fe p = Fr::from_mont(mont);
    memcpy(dst, &p, 32);
}

}  // namespace wsnark

using namespace wsnark;

struct wsnark_synth { SynthCircuit c; };

extern "C" {

int wsnark_synth_new(uint32_t log_domain, uint32_t n_public, uint64_t circuit_seed, uint64_t setup_seed, int style,
                     wsnark_synth_t** out) {
    if (!out) return WSNARK_ERR_ARG;
    wsnark_synth* h = new (std::nothrow) wsnark_synth();
    if (!h) return WSNARK_ERR_ARG;
    const int rc = synth_build(log_domain, n_public, circuit_seed, setup_seed, style, &h->c);
    if (rc) { delete h; return rc; }
    *out = h;
    return WSNARK_OK;
}

void wsnark_synth_free(wsnark_synth_t* h) { delete h; }

static uint64_t pols_len(const Sparse& M, uint32_t n_vars) { return (uint64_t)n_vars * 4 + (uint64_t)M.row.size() * 36; }

int wsnark_synth_info(const wsnark_synth_t* h, wsnark_synth_info_t* o) {
    if (!h || !o) return WSNARK_ERR_ARG;
    const SynthCircuit& S = h->c;
    o->n_vars = S.n_vars; o->n_public = S.n_public; o->domain = S.domain;
    o->nnz_a = S.A.row.size(); o->nnz_b = S.B.row.size();
    o->absent_a = S.A.absent; o->absent_b = S.B.absent;
    o->pols_a_len = pols_len(S.A, S.n_vars); o->pols_b_len = pols_len(S.B, S.n_vars);
    o->n_g1_scalars = 3 + 2 * (uint64_t)S.n_vars + ((uint64_t)S.n_vars - S.n_public - 1) + S.domain + S.n_public + 1;
    o->n_g2_scalars = 3 + (uint64_t)S.n_vars;
    return WSNARK_OK;
}

int wsnark_synth_witness(const wsnark_synth_t* h, void* out) {
    if (!h || !out) return WSNARK_ERR_ARG;
    const SynthCircuit& S = h->c;
    uint8_t* o = (uint8_t*)out;
    parallel_for(S.n_vars, [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) store_plain_fr(o + i * 32, S.w[i]); });
    return WSNARK_OK;
}

/* tools/buildpkey.js:79-89 writeTransformedPolynomial: per signal u32 count, then (u32 idx, 32 B coef Montgomery) */
int wsnark_synth_pols(const wsnark_synth_t* h, int which, void* out, uint64_t cap) {
    if (!h || !out || (which != 0 && which != 1)) return WSNARK_ERR_ARG;
    const SynthCircuit& S = h->c;
    const Sparse& M = which ? S.B : S.A;
    if (cap < pols_len(M, S.n_vars)) return WSNARK_ERR_SIZE;
    uint8_t* o = (uint8_t*)out;
    parallel_for(S.n_vars, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t s = lo; s < hi; s++) {
            uint8_t* p = o + s * 4 + M.col_ptr[s] * 36;
            const uint32_t cnt = (uint32_t)(M.col_ptr[s + 1] - M.col_ptr[s]);
            memcpy(p, &cnt, 4); p += 4;
            for (uint64_t k = M.col_ptr[s]; k < M.col_ptr[s + 1]; k++) { memcpy(p, &M.row[k], 4); memcpy(p + 4, &M.coef[k], 32); p += 36; }
        }
    });
    return WSNARK_OK;
}

/* group 1: alfa1, beta1, delta1, A[nVars], B1[nVars], C[nVars - nPublic - 1] (signals nPublic+1..), hExps[domain],
 * IC[nPublic + 1]; group 2: beta2, delta2, gamma2, B2[nVars].  32-byte plain little-endian each. */
int wsnark_synth_key_scalars(const wsnark_synth_t* h, int group, void* out) {
    if (!h || !out || (group != 1 && group != 2)) return WSNARK_ERR_ARG;
    const SynthCircuit& S = h->c;
    uint8_t* o = (uint8_t*)out;
    const uint32_t nv = S.n_vars, np = S.n_public, dom = S.domain;
    if (group == 2) {
        store_plain_fr(o, S.beta); store_plain_fr(o + 32, S.delta); store_plain_fr(o + 64, S.gamma);
        parallel_for(nv, [&](uint64_t lo, uint64_t hi) { for (uint64_t s = lo; s < hi; s++) store_plain_fr(o + 96 + s * 32, S.b[s]); });
        return WSNARK_OK;
    }
    const Fe dinv = Fr::inv(S.delta), ginv = Fr::inv(S.gamma);
    store_plain_fr(o, S.alpha); store_plain_fr(o + 32, S.beta); store_plain_fr(o + 64, S.delta);
    uint8_t* pa = o + 96;
    uint8_t* pb = pa + (uint64_t)nv * 32;
    uint8_t* pc = pb + (uint64_t)nv * 32;
    uint8_t* ph = pc + ((uint64_t)nv - np - 1) * 32;
    uint8_t* pic = ph + (uint64_t)dom * 32;
    parallel_for(nv, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t s = lo; s < hi; s++) {
            store_plain_fr(pa + s * 32, S.a[s]);
            store_plain_fr(pb + s * 32, S.b[s]);
            const Fe kc = Fr::add(Fr::add(Fr::mul(S.beta, S.a[s]), Fr::mul(S.alpha, S.b[s])), S.c[s]);
            if (s > np) store_plain_fr(pc + (s - np - 1) * 32, Fr::mul(kc, dinv));
            else store_plain_fr(pic + s * 32, Fr::mul(kc, ginv));
        }
    });
    const Fe zd = Fr::mul(S.z, dinv);
    parallel_for(dom, [&](uint64_t lo, uint64_t hi) {                      // hExps[i] = tau^i Z(tau) / delta
        Fe t = Fr::mul(zd, Fr::pow_u64(S.tau, lo));
        for (uint64_t i = lo; i < hi; i++) { store_plain_fr(ph + i * 32, t); t = Fr::mul(t, S.tau); }
    });
    return WSNARK_OK;
}

/* discrete logarithms (a | b | c, 32-byte plain LE each) of pi_a, pi_b, pi_c w.r.t. the G1 / G2 / G1 generators for
 * the blinding bytes r32, s32 (raw 256-bit, reduced mod r like src/bn128.js:642-661 implies for points of order r) */
int wsnark_synth_expected(const wsnark_synth_t* h, const void* r32, const void* s32, void* out96) {
    if (!h || !r32 || !s32 || !out96) return WSNARK_ERR_ARG;
    const SynthCircuit& S = h->c;
    Fe r, s;
    memcpy(&r, r32, 32); memcpy(&s, s32, 32);
    r = Fr::to_mont(Fr::reduce_full(r)); s = Fr::to_mont(Fr::reduce_full(s));
    Fe Aw = Fr::zero(), Bw = Fr::zero(), Cw = Fr::zero(), priv = Fr::zero();
    for (uint32_t i = 0; i < S.n_vars; i++) {
        Aw = Fr::add(Aw, Fr::mul(S.w[i], S.a[i]));
        Bw = Fr::add(Bw, Fr::mul(S.w[i], S.b[i]));
        Cw = Fr::add(Cw, Fr::mul(S.w[i], S.c[i]));
        if (i > S.n_public) {
            const Fe kc = Fr::add(Fr::add(Fr::mul(S.beta, S.a[i]), Fr::mul(S.alpha, S.b[i])), S.c[i]);
            priv = Fr::add(priv, Fr::mul(S.w[i], kc));
        }
    }
    const Fe dinv = Fr::inv(S.delta);
    const Fe hz = Fr::sub(Fr::mul(Aw, Bw), Cw);                            // h(tau) Z(tau) = A(tau) B(tau) - C(tau)
    const Fe a = Fr::add(Fr::add(S.alpha, Aw), Fr::mul(r, S.delta));
    const Fe b = Fr::add(Fr::add(S.beta, Bw), Fr::mul(s, S.delta));
    Fe c = Fr::mul(Fr::add(priv, hz), dinv);
    c = Fr::add(c, Fr::mul(s, a));
    c = Fr::add(c, Fr::mul(r, b));
    c = Fr::sub(c, Fr::mul(Fr::mul(r, s), S.delta));
    uint8_t* o = (uint8_t*)out96;
    store_plain_fr(o, a); store_plain_fr(o + 32, b); store_plain_fr(o + 64, c);
    return WSNARK_OK;
}

}  // extern "C"
