/*
 * bn128_oracle.c -- CPU restatement (plain C) of wasmsnark's BN128 Groth16
 * prove hot path.  TEST INFRASTRUCTURE ONLY (see bn128_oracle.h).
 * Parity status: PINNED against reference-generated vectors in tests/golden/.
 */
#include "bn128_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef orc_u256 u256;

/* ------------------------------------------------------------------ */
/* constants: src/bn128/build_bn128.js:19-20 (q, r)                     */
/* ------------------------------------------------------------------ */
typedef struct {
    u256 p;        /* modulus */
    uint64_t np;   /* -p^-1 mod 2^64   (build_f1m.js:255 uses the 32-bit truncation) */
    u256 R;        /* 2^256 mod p      (build_f1m.js:37-38) */
    u256 R2;       /* 2^512 mod p */
    u256 pm2;      /* p - 2 (Fermat exponent) */
} field_t;

static field_t FLD[2];
static int g_init_done = 0;

static const u256 MOD_Q = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static const u256 MOD_R = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};

/* ---- build_int.js:190-283 add/sub with carry; :152-188 gte ---- */
static uint64_t int_add(const u256 *a, const u256 *b, u256 *r) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t int_sub(const u256 *a, const u256 *b, u256 *r) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        r->l[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static int int_gte(const u256 *a, const u256 *b) {
    for (int i = 3; i >= 0; i--) { if (a->l[i] > b->l[i]) return 1; if (a->l[i] < b->l[i]) return 0; }
    return 1;
}
static int int_is_zero(const u256 *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int int_eq(const u256 *a, const u256 *b) {
    return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3];
}

/* ---- build_f1m.js:67-113  add/sub/neg with conditional correction ---- */
static void f_add(const field_t *F, const u256 *a, const u256 *b, u256 *r) {
    u256 t; uint64_t c = int_add(a, b, &t);
    if (c || int_gte(&t, &F->p)) int_sub(&t, &F->p, &t);
    *r = t;
}
static void f_sub(const field_t *F, const u256 *a, const u256 *b, u256 *r) {
    u256 t; if (int_sub(a, b, &t)) int_add(&t, &F->p, &t);
    *r = t;
}
static void f_neg(const field_t *F, const u256 *a, u256 *r) {
    if (int_is_zero(a)) { *r = *a; return; }
    int_sub(&F->p, a, r);
}

/* ---- build_f1m.js:235-436  Montgomery product x*y*2^-256 mod p, canonical.
 * The reference scans products over 8x32-bit limbs; this is the same map
 * computed with 4x64-bit CIOS (the output is the unique canonical value). ---- */
static void f_mul(const field_t *F, const u256 *a, const u256 *b, u256 *r) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->np;
        c = (u128)m * F->p.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F->p.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    u256 o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || int_gte(&o, &F->p)) int_sub(&o, &F->p, &o);
    *r = o;
}
static void f_sq(const field_t *F, const u256 *a, u256 *r) { f_mul(F, a, a, r); } /* build_f1m.js:439-736 */

/* build_f1m.js:749-770 */
static void f_to_mont(const field_t *F, const u256 *a, u256 *r) { f_mul(F, a, &F->R2, r); }
static void f_from_mont(const field_t *F, const u256 *a, u256 *r) {
    u256 one = {{1, 0, 0, 0}}; f_mul(F, a, &one, r);
}
/* build_f1m.js:772-782 (the reference uses ext-Euclid on the plain value; the
 * inverse is unique, so Fermat gives the same canonical output) */
static void f_inv(const field_t *F, const u256 *a, u256 *r) {
    u256 acc = F->R, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((F->pm2.l[i >> 6] >> (i & 63)) & 1) f_mul(F, &acc, &base, &acc);
        f_mul(F, &base, &base, &base);
    }
    *r = acc;
}

static void field_setup(field_t *F, const u256 *p) {
    F->p = *p;
    uint64_t inv = 1;                       /* Newton: inv = p^-1 mod 2^64 */
    for (int i = 0; i < 6; i++) inv *= 2 - p->l[0] * inv;
    F->np = (uint64_t)0 - inv;
    u256 x = {{1, 0, 0, 0}};                /* 2^k mod p by doubling */
    for (int i = 0; i < 512; i++) {
        u256 t; uint64_t c = int_add(&x, &x, &t);
        if (c || int_gte(&t, p)) int_sub(&t, p, &t);
        x = t;
        if (i == 255) F->R = x;
    }
    F->R2 = x;
    u256 two = {{2, 0, 0, 0}};
    int_sub(p, &two, &F->pm2);
}

/* ---- FFT tables: build_fft.js:29-72 ---- */
static u256 ROOTS[29];   /* ROOTS[s] = w_{2^s} in Montgomery form */
static u256 INV2[29];    /* INV2[s] = 2^-s in Montgomery form */

static void f_pow(const field_t *F, const u256 *base_m, const u256 *e, u256 *r) {
    u256 acc = F->R, b = *base_m;
    for (int i = 0; i < 256; i++) {
        if ((e->l[i >> 6] >> (i & 63)) & 1) f_mul(F, &acc, &b, &acc);
        f_mul(F, &b, &b, &b);
    }
    *r = acc;
}

void orc_init(void) {
    if (g_init_done) return;
    field_setup(&FLD[0], &MOD_Q);
    field_setup(&FLD[1], &MOD_R);
    const field_t *F = &FLD[1];
    /* build_fft.js:29-47: rem = (r-1) >> 28; nr = 5 (smallest non-residue);
     * w[28] = nr^rem; w[n] = w[n+1]^2 */
    u256 rem = F->p; rem.l[0] -= 1;
    for (int i = 0; i < 28; i++) {          /* rem >>= 1 */
        for (int k = 0; k < 4; k++) rem.l[k] = (rem.l[k] >> 1) | (k < 3 ? rem.l[k + 1] << 63 : 0);
    }
    u256 five = {{5, 0, 0, 0}}, five_m;
    f_to_mont(F, &five, &five_m);
    f_pow(F, &five_m, &rem, &ROOTS[28]);
    for (int n = 27; n >= 0; n--) f_mul(F, &ROOTS[n + 1], &ROOTS[n + 1], &ROOTS[n]);
    /* build_fft.js:59-72: INV2[i] = (2^i)^-1 */
    u256 two = {{2, 0, 0, 0}}, two_m, half;
    f_to_mont(F, &two, &two_m);
    f_inv(F, &two_m, &half);
    INV2[0] = F->R;
    for (int i = 1; i <= 28; i++) f_mul(F, &INV2[i - 1], &half, &INV2[i]);
    g_init_done = 1;
}

/* ---- public field API ---- */
void orc_f_mul(int w, const u256 *a, const u256 *b, u256 *r) { orc_init(); f_mul(&FLD[w], a, b, r); }
void orc_f_square(int w, const u256 *a, u256 *r) { orc_init(); f_sq(&FLD[w], a, r); }
void orc_f_add(int w, const u256 *a, const u256 *b, u256 *r) { orc_init(); f_add(&FLD[w], a, b, r); }
void orc_f_sub(int w, const u256 *a, const u256 *b, u256 *r) { orc_init(); f_sub(&FLD[w], a, b, r); }
void orc_f_neg(int w, const u256 *a, u256 *r) { orc_init(); f_neg(&FLD[w], a, r); }
void orc_f_to_mont(int w, const u256 *a, u256 *r) { orc_init(); f_to_mont(&FLD[w], a, r); }
void orc_f_from_mont(int w, const u256 *a, u256 *r) { orc_init(); f_from_mont(&FLD[w], a, r); }
void orc_f_inverse(int w, const u256 *a, u256 *r) { orc_init(); f_inv(&FLD[w], a, r); }
void orc_f_constants(int w, u256 *m, u256 *R, u256 *R2, uint64_t *np) {
    orc_init(); *m = FLD[w].p; *R = FLD[w].R; *R2 = FLD[w].R2; *np = FLD[w].np;
}

/* ------------------------------------------------------------------ */
/* Fq2 = Fq[u]/(u^2+1): build_f2m.js (non-residue fn = f1m_neg,         */
/* build_bn128.js:40)                                                   */
/* ------------------------------------------------------------------ */
#define FQ (&FLD[0])
static void f2_add(const orc_fq2 *a, const orc_fq2 *b, orc_fq2 *r) { f_add(FQ, &a->c0, &b->c0, &r->c0); f_add(FQ, &a->c1, &b->c1, &r->c1); }
static void f2_sub(const orc_fq2 *a, const orc_fq2 *b, orc_fq2 *r) { f_sub(FQ, &a->c0, &b->c0, &r->c0); f_sub(FQ, &a->c1, &b->c1, &r->c1); }
static void f2_neg(const orc_fq2 *a, orc_fq2 *r) { f_neg(FQ, &a->c0, &r->c0); f_neg(FQ, &a->c1, &r->c1); }
/* build_f2m.js:127-163  Karatsuba: A=a0b0, B=a1b1, C=(a0+a1)(b0+b1); r0=A+nr(B), r1=C-(A+B) */
static void f2_mul(const orc_fq2 *a, const orc_fq2 *b, orc_fq2 *r) {
    u256 A, B, C, D, t0, t1;
    f_mul(FQ, &a->c0, &b->c0, &A);
    f_mul(FQ, &a->c1, &b->c1, &B);
    f_add(FQ, &a->c0, &a->c1, &t0);
    f_add(FQ, &b->c0, &b->c1, &t1);
    f_mul(FQ, &t0, &t1, &C);
    f_add(FQ, &A, &B, &D);
    f_sub(FQ, &A, &B, &r->c0);
    f_sub(FQ, &C, &D, &r->c1);
}
/* build_f2m.js:186-227  complex squaring: AB=a0a1; r0=(a0+a1)(a0+nr a1)-AB-nr AB; r1=2AB */
static void f2_sq(const orc_fq2 *a, orc_fq2 *r) {
    u256 AB, s, d, t;
    f_mul(FQ, &a->c0, &a->c1, &AB);
    f_add(FQ, &a->c0, &a->c1, &s);
    f_sub(FQ, &a->c0, &a->c1, &d);
    f_mul(FQ, &s, &d, &t);
    r->c0 = t;
    f_add(FQ, &AB, &AB, &r->c1);
}
/* build_f2m.js:353-383  inverse via norm: t = (a0^2 - nr a1^2)^-1; r = (a0 t, -a1 t) */
static void f2_inv(const orc_fq2 *a, orc_fq2 *r) {
    u256 t0, t1, t;
    f_sq(FQ, &a->c0, &t0);
    f_sq(FQ, &a->c1, &t1);
    f_add(FQ, &t0, &t1, &t);
    f_inv(FQ, &t, &t);
    f_mul(FQ, &a->c0, &t, &r->c0);
    f_mul(FQ, &a->c1, &t, &t1);
    f_neg(FQ, &t1, &r->c1);
}
static int f2_is_zero(const orc_fq2 *a) { return int_is_zero(&a->c0) && int_is_zero(&a->c1); }
static int f2_eq(const orc_fq2 *a, const orc_fq2 *b) { return int_eq(&a->c0, &b->c0) && int_eq(&a->c1, &b->c1); }
static void f2_from_mont(const orc_fq2 *a, orc_fq2 *r) { f_from_mont(FQ, &a->c0, &r->c0); f_from_mont(FQ, &a->c1, &r->c1); }

void orc_f2_mul(const orc_fq2 *a, const orc_fq2 *b, orc_fq2 *r) { orc_init(); orc_fq2 t; f2_mul(a, b, &t); *r = t; }
void orc_f2_square(const orc_fq2 *a, orc_fq2 *r) { orc_init(); orc_fq2 t; f2_sq(a, &t); *r = t; }
void orc_f2_inverse(const orc_fq2 *a, orc_fq2 *r) { orc_init(); orc_fq2 t; f2_inv(a, &t); *r = t; }

/* ------------------------------------------------------------------ */
/* G1 over Fq                                                           */
/* ------------------------------------------------------------------ */
static void q_mul(const u256 *a, const u256 *b, u256 *r) { f_mul(FQ, a, b, r); }
static void q_sq(const u256 *a, u256 *r) { f_sq(FQ, a, r); }
static void q_add(const u256 *a, const u256 *b, u256 *r) { f_add(FQ, a, b, r); }
static void q_sub(const u256 *a, const u256 *b, u256 *r) { f_sub(FQ, a, b, r); }
static void q_neg(const u256 *a, u256 *r) { f_neg(FQ, a, r); }
static void q_inv(const u256 *a, u256 *r) { f_inv(FQ, a, r); }
static void q_set_zero(u256 *r) { memset(r, 0, sizeof *r); }
static void q_set_one(u256 *r) { *r = FQ->R; }
static void q_from_mont(const u256 *a, u256 *r) { f_from_mont(FQ, a, r); }

#define FE u256
#define PT orc_g1
#define G(n) orc_g1_##n
#define AFF_BYTES 64
#define FE_mul q_mul
#define FE_sq q_sq
#define FE_add q_add
#define FE_sub q_sub
#define FE_neg q_neg
#define FE_inv q_inv
#define FE_is_zero int_is_zero
#define FE_eq int_eq
#define FE_set_zero q_set_zero
#define FE_set_one q_set_one
#define FE_from_mont q_from_mont
#include "curve_tmpl.inc"
#undef FE
#undef PT
#undef G
#undef AFF_BYTES
#undef FE_mul
#undef FE_sq
#undef FE_add
#undef FE_sub
#undef FE_neg
#undef FE_inv
#undef FE_is_zero
#undef FE_eq
#undef FE_set_zero
#undef FE_set_one
#undef FE_from_mont

/* ------------------------------------------------------------------ */
/* G2 over Fq2                                                          */
/* ------------------------------------------------------------------ */
static void f2_set_zero(orc_fq2 *r) { memset(r, 0, sizeof *r); }
static void f2_set_one(orc_fq2 *r) { r->c0 = FQ->R; memset(&r->c1, 0, sizeof r->c1); }
static void f2_mul_alias(const orc_fq2 *a, const orc_fq2 *b, orc_fq2 *r) { orc_fq2 t; f2_mul(a, b, &t); *r = t; }
static void f2_sq_alias(const orc_fq2 *a, orc_fq2 *r) { orc_fq2 t; f2_sq(a, &t); *r = t; }
static void f2_inv_alias(const orc_fq2 *a, orc_fq2 *r) { orc_fq2 t; f2_inv(a, &t); *r = t; }

#define FE orc_fq2
#define PT orc_g2
#define G(n) orc_g2_##n
#define AFF_BYTES 128
#define FE_mul f2_mul_alias
#define FE_sq f2_sq_alias
#define FE_add f2_add
#define FE_sub f2_sub
#define FE_neg f2_neg
#define FE_inv f2_inv_alias
#define FE_is_zero f2_is_zero
#define FE_eq f2_eq
#define FE_set_zero f2_set_zero
#define FE_set_one f2_set_one
#define FE_from_mont f2_from_mont
#include "curve_tmpl.inc"
#undef FE
#undef PT
#undef G

/* ------------------------------------------------------------------ */
/* host-level MSM sharding: src/bn128.js:353-383 (G1), :385-415 (G2)    */
/* floor(n/W) pairs per worker, remainder to the last, then a serial    */
/* EC sum of the W partials on the main thread.                         */
/* ------------------------------------------------------------------ */
void orc_g1_multiexp_workers(const uint8_t *scalars, const uint8_t *points, uint32_t n, int workers, orc_g1 *r) {
    orc_init();
    if (workers < 1) workers = 1;
    orc_g1 *part = (orc_g1 *)malloc((size_t)workers * sizeof(orc_g1));
    uint32_t per = n / (uint32_t)workers;
#pragma omp parallel for schedule(static, 1) num_threads(workers)
    for (int i = 0; i < workers; i++) {
        uint32_t cnt = (i < workers - 1) ? per : n - per * (uint32_t)(workers - 1);
        orc_g1_zero(&part[i]);                                  /* worker: g1m_zero(pRes) bn128.js:108 */
        orc_g1_multiexp2(scalars + (size_t)i * per * 32, points + (size_t)i * per * 64, cnt, 7, &part[i]);
    }
    orc_g1 acc; orc_g1_zero(&acc);
    for (int i = 0; i < workers; i++) orc_g1_add(&acc, &part[i], &acc);
    *r = acc;
    free(part);
}
void orc_g2_multiexp_workers(const uint8_t *scalars, const uint8_t *points, uint32_t n, int workers, orc_g2 *r) {
    orc_init();
    if (workers < 1) workers = 1;
    orc_g2 *part = (orc_g2 *)malloc((size_t)workers * sizeof(orc_g2));
    uint32_t per = n / (uint32_t)workers;
#pragma omp parallel for schedule(static, 1) num_threads(workers)
    for (int i = 0; i < workers; i++) {
        uint32_t cnt = (i < workers - 1) ? per : n - per * (uint32_t)(workers - 1);
        orc_g2_zero(&part[i]);
        orc_g2_multiexp(scalars + (size_t)i * per * 32, points + (size_t)i * per * 128, cnt, 7, &part[i]);
    }
    orc_g2 acc; orc_g2_zero(&acc);
    for (int i = 0; i < workers; i++) orc_g2_add(&acc, &part[i], &acc);
    *r = acc;
    free(part);
}

/* ------------------------------------------------------------------ */
/* FFT: src/build_fft.js                                                */
/* ------------------------------------------------------------------ */
#define FR (&FLD[1])

/* build_fft.js:92-157 __log2: traps unless n is a power of two <= 2^28 */
static int fft_log2(uint32_t n) {
    if (n == 0 || (n & (n - 1))) return -1;
    int b = 0; while ((1u << b) < n) b++;
    return b > 28 ? -1 : b;
}
/* build_fft.js:717-786 __rev + :650-715 __reversePermutation */
static void reverse_permutation(u256 *x, int bits) {
    uint32_t n = 1u << bits;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r = 0;
        for (int b = 0; b < bits; b++) if (i & (1u << b)) r |= 1u << (bits - 1 - b);
        if (i < r) { u256 t = x[i]; x[i] = x[r]; x[r] = t; }
    }
}
/* build_fft.js:223-372 __rawfft: bit-reverse, then DIT stages s=1..bits with
 * m=2^s; per block k the twiddle run starts at 1 (odd=0) or w_{2m} (odd=1)
 * and is multiplied by w_m each butterfly (:276-287, :326-358) */
static void rawfft(u256 *x, int bits, int odd) {
    uint32_t n = 1u << bits;
    reverse_permutation(x, bits);
    for (int s = 1; s <= bits; s++) {
        uint32_t m = 1u << s, mdiv2 = m >> 1;
        const u256 *pwm = &ROOTS[s];
        for (uint32_t k = 0; k < n; k += m) {
            u256 W = odd ? ROOTS[s + 1] : FR->R;
            for (uint32_t j = 0; j < mdiv2; j++) {
                u256 T, U;
                f_mul(FR, &W, &x[k + j + mdiv2], &T);
                U = x[k + j];
                f_add(FR, &U, &T, &x[k + j]);
                f_sub(FR, &U, &T, &x[k + j + mdiv2]);
                f_mul(FR, &W, pwm, &W);
            }
        }
    }
}
/* build_fft.js:550-648 __finalInverse: swap i <-> n-i and scale by 2^-bits */
static void final_inverse(u256 *x, int bits) {
    uint32_t n = 1u << bits, ndiv2 = n >> 1;
    const u256 *inv = &INV2[bits];
    for (uint32_t i = 1; i < ndiv2; i++) {
        u256 T = x[i];
        f_mul(FR, &x[n - i], inv, &x[i]);
        f_mul(FR, &T, inv, &x[n - i]);
    }
    f_mul(FR, &x[0], inv, &x[0]);
    f_mul(FR, &x[ndiv2], inv, &x[ndiv2]);
}
int orc_fft(u256 *x, uint32_t n, int odd) {   /* build_fft.js:159-187 */
    orc_init();
    int bits = fft_log2(n); if (bits < 0) return -1;
    if (odd && bits >= 28) return -1;           /* ROOTS[29] does not exist */
    rawfft(x, bits, odd);
    return 0;
}
int orc_ifft(u256 *x, uint32_t n, int odd) {  /* build_fft.js:189-221 */
    orc_init();
    int bits = fft_log2(n); if (bits < 0) return -1;
    if (odd && bits >= 28) return -1;
    if (bits == 0) return -1;   /* reference: __finalInverse(n=1) loops out of bounds (build_fft.js:575-583) */
    rawfft(x, bits, odd);
    final_inverse(x, bits);
    return 0;
}
void orc_fr_to_mont_n(const u256 *in, u256 *out, uint32_t n) {   /* build_fft.js:418-458 */
    orc_init(); for (uint32_t i = 0; i < n; i++) f_to_mont(FR, &in[i], &out[i]);
}
void orc_fr_from_mont_n(const u256 *in, u256 *out, uint32_t n) { /* build_fft.js:507-547 */
    orc_init(); for (uint32_t i = 0; i < n; i++) f_from_mont(FR, &in[i], &out[i]);
}

/* ------------------------------------------------------------------ */
/* build_pol.js:62-144 pol_constructLC: per signal: u32 ncoefs, then    */
/* ncoefs x (u32 idx, 32 B coef);  res[idx] += signal * coef            */
/* ------------------------------------------------------------------ */
int64_t orc_pol_construct_lc(const uint8_t *pols, size_t pols_len, const u256 *sig, uint32_t n_signals,
                             u256 *res, uint32_t domain) {
    orc_init();
    size_t pp = 0;
    for (uint32_t i = 0; i < n_signals; i++) {
        uint32_t ncoefs;
        if (pp + 4 > pols_len) return -1;
        memcpy(&ncoefs, pols + pp, 4); pp += 4;
        for (uint32_t j = 0; j < ncoefs; j++) {
            uint32_t idx; u256 coef, aux;
            if (pp + 36 > pols_len) return -1;
            memcpy(&idx, pols + pp, 4); pp += 4;
            memcpy(&coef, pols + pp, 32); pp += 32;
            if (idx >= domain) return -1;
            f_mul(FR, &sig[i], &coef, &aux);
            f_add(FR, &aux, &res[idx], &res[idx]);
        }
    }
    return (int64_t)pp;
}

/* ------------------------------------------------------------------ */
/* worker command CALC_H: src/bn128.js:126-166                          */
/* ------------------------------------------------------------------ */
int orc_calc_h(const uint8_t *signals, const uint8_t *polsA, size_t lenA, const uint8_t *polsB, size_t lenB,
               uint32_t n_signals, uint32_t domain, uint8_t *out_h) {
    orc_init();
    if (fft_log2(domain) < 0 || fft_log2(domain) >= 28) return -1;
    size_t D = domain;
    u256 *sigM = (u256 *)malloc((size_t)n_signals * 32 + 32);
    u256 *A = (u256 *)calloc(D, 32), *B = (u256 *)calloc(D, 32);      /* pol_zero :141-142 */
    u256 *A2 = (u256 *)malloc(D * 64), *B2 = (u256 *)malloc(D * 64);
    int rc = 0;
    orc_fr_to_mont_n((const u256 *)signals, sigM, n_signals);          /* :139 */
    if (orc_pol_construct_lc(polsA, lenA, sigM, n_signals, A, domain) < 0) rc = -2;   /* :144 */
    if (orc_pol_construct_lc(polsB, lenB, sigM, n_signals, B, domain) < 0) rc = -2;   /* :145 */
    if (rc == 0) {
        for (size_t i = 0; i < D; i++) { A2[2 * i] = A[i]; B2[2 * i] = B[i]; }        /* :147-148 */
        orc_ifft(A, domain, 0); orc_ifft(B, domain, 0);                               /* :150-151 */
        orc_fft(A, domain, 1);  orc_fft(B, domain, 1);                                /* :152-153 */
        for (size_t i = 0; i < D; i++) { A2[2 * i + 1] = A[i]; B2[2 * i + 1] = B[i]; }/* :155-156 */
        for (size_t i = 0; i < 2 * D; i++) f_mul(FR, &A2[i], &B2[i], &A2[i]);         /* :158 */
        orc_ifft(A2, domain * 2, 0);                                                  /* :160 */
        orc_fr_from_mont_n(A2 + D, (u256 *)out_h, domain);                            /* :162-164 */
    }
    free(sigM); free(A); free(B); free(A2); free(B2);
    return rc;
}

/* ------------------------------------------------------------------ */
/* Bn128.groth16GenProof: src/bn128.js:580-720                          */
/* ------------------------------------------------------------------ */
static void load_point1(const uint8_t *b, orc_g1 *p) { /* bn128.js:441-446: append z = 1 */
    memcpy(&p->x, b, 32); memcpy(&p->y, b + 32, 32); p->z = FQ->R;
}
static void load_point2(const uint8_t *b, orc_g2 *p) { /* bn128.js:448-453 */
    memcpy(&p->x, b, 64); memcpy(&p->y, b + 64, 64); p->z.c0 = FQ->R; memset(&p->z.c1, 0, 32);
}
/* build_int.js:285-580 int_mul: 256x256 -> 512-bit schoolbook */
static void int_mul_512(const uint8_t *a32, const uint8_t *b32, uint8_t out64[64]) {
    uint64_t a[4], b[4], t[8] = {0};
    memcpy(a, a32, 32); memcpy(b, b32, 32);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[i] * b[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        t[i + 4] = (uint64_t)c;
    }
    memcpy(out64, t, 64);
}

int orc_groth16_prove(const uint8_t *witness, size_t witness_len, const uint8_t *pkey, size_t pkey_len,
                      const uint8_t *r32, const uint8_t *s32, int workers, uint8_t *out384) {
    orc_init();
    if (pkey_len < 40 + 448) return -1;
    uint32_t h[10]; memcpy(h, pkey, 40);                                   /* bn128.js:581-591 */
    uint32_t nSignals = h[0], nPublic = h[1], domain = h[2];
    uint32_t pPolsA = h[3], pPolsB = h[4], pA = h[5], pB1 = h[6], pB2 = h[7], pC = h[8], pH = h[9];
    if (witness_len < (size_t)nSignals * 32) return -1;
    if ((size_t)pH + (size_t)domain * 64 > pkey_len) return -1;
    if (nPublic + 1 > nSignals) return -1;

    /* CALC_H then H MSM (:607-615) */
    uint8_t *hcoef = (uint8_t *)malloc((size_t)domain * 32);
    int rc = orc_calc_h(witness, pkey + pPolsA, pPolsB - pPolsA, pkey + pPolsB, pA - pPolsB, nSignals, domain, hcoef);
    if (rc) { free(hcoef); return rc; }
    orc_g1 sA, sB1, sC, sH; orc_g2 sB2;
    orc_g1_multiexp_workers(hcoef, pkey + pH, domain, workers, &sH);               /* :614 */
    orc_g1_multiexp_workers(witness, pkey + pA, nSignals, workers, &sA);           /* :617 */
    orc_g1_multiexp_workers(witness, pkey + pB1, nSignals, workers, &sB1);         /* :618 */
    orc_g2_multiexp_workers(witness, pkey + pB2, nSignals, workers, &sB2);         /* :619 */
    orc_g1_multiexp_workers(witness + (size_t)(nPublic + 1) * 32, pkey + pC, nSignals - nPublic - 1, workers, &sC); /* :620 */
    free(hcoef);

    orc_g1 alfa1, beta1, delta1, aux1, pi_a = sA, pib1 = sB1, pi_c = sC;
    orc_g2 beta2, delta2, aux2, pi_b = sB2;
    load_point1(pkey + 40, &alfa1); load_point1(pkey + 40 + 64, &beta1); load_point1(pkey + 40 + 128, &delta1);
    load_point2(pkey + 40 + 192, &beta2); load_point2(pkey + 40 + 320, &delta2);   /* :599-603, 633-637 */

    orc_g1_add(&alfa1, &pi_a, &pi_a);                          /* :671-673 */
    orc_g1_times_scalar(&delta1, r32, 32, &aux1);
    orc_g1_add(&aux1, &pi_a, &pi_a);
    orc_g2_add(&beta2, &pi_b, &pi_b);                          /* :676-678 */
    orc_g2_times_scalar(&delta2, s32, 32, &aux2);
    orc_g2_add(&aux2, &pi_b, &pi_b);
    orc_g1_add(&beta1, &pib1, &pib1);                          /* :681-683 */
    orc_g1_times_scalar(&delta1, s32, 32, &aux1);
    orc_g1_add(&aux1, &pib1, &pib1);
    orc_g1_add(&sH, &pi_c, &pi_c);                             /* :687-688 */
    orc_g1_times_scalar(&pi_a, s32, 32, &aux1);                /* :692-693 */
    orc_g1_add(&aux1, &pi_c, &pi_c);
    orc_g1_times_scalar(&pib1, r32, 32, &aux1);                /* :696-697 */
    orc_g1_add(&aux1, &pi_c, &pi_c);
    uint8_t prs[64];
    int_mul_512(r32, s32, prs);                                /* :700-701 */
    orc_g1_times_scalar(&delta1, prs, 64, &aux1);              /* :702 */
    orc_g1_neg(&aux1, &aux1);
    orc_g1_add(&aux1, &pi_c, &pi_c);

    orc_g1_affine(&pi_a, &pi_a); orc_g2_affine(&pi_b, &pi_b); orc_g1_affine(&pi_c, &pi_c);          /* :706-708 */
    orc_g1_from_mont(&pi_a, &pi_a); orc_g2_from_mont(&pi_b, &pi_b); orc_g1_from_mont(&pi_c, &pi_c); /* :710-712 */
    memcpy(out384, &pi_a, 96); memcpy(out384 + 96, &pi_b, 192); memcpy(out384 + 288, &pi_c, 96);
    return 0;
}
