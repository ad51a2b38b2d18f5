/*
 * bn128_oracle.h -- CPU restatement (plain C) of the BN128 Groth16 prove hot
 * path of iden3/wasmsnark.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (wasmsnark_amd/csrc -> libwsnark.so) never
 * links, calls or falls back to it.
 *
 * Parity status: PINNED.  The reference publishes no golden vectors for this
 * path, so the oracle is pinned against outputs of the
 * reference itself (its prebuilt WASM module run under Node in the build
 * container by oracle/ref_harness/gen_golden.js); the outputs are committed
 * under tests/golden/ and checked by tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference root).
 *
 * Data layouts are the reference's: field elements are 32-byte little-endian
 * (8 x u32 in the reference == 4 x u64 here), Montgomery form with R = 2^256
 * unless a name says "plain".  G1 affine = 64 B (x,y); G2 affine = 128 B
 * (x.c0,x.c1,y.c0,y.c1); Jacobian = 96 B / 192 B (x,y,z).
 */
#ifndef BN128_ORACLE_H
#define BN128_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } orc_u256;
typedef struct { orc_u256 c0, c1; } orc_fq2;
typedef struct { orc_u256 x, y, z; } orc_g1;     /* Jacobian, Montgomery */
typedef struct { orc_fq2 x, y, z; } orc_g2;

/* one-time constant setup (idempotent, called lazily by everything) */
void orc_init(void);

/* ---- field: which = 0 -> Fq (base field), 1 -> Fr (scalar field) ---- */
void orc_f_mul(int which, const orc_u256 *a, const orc_u256 *b, orc_u256 *r);
void orc_f_square(int which, const orc_u256 *a, orc_u256 *r);
void orc_f_add(int which, const orc_u256 *a, const orc_u256 *b, orc_u256 *r);
void orc_f_sub(int which, const orc_u256 *a, const orc_u256 *b, orc_u256 *r);
void orc_f_neg(int which, const orc_u256 *a, orc_u256 *r);
void orc_f_to_mont(int which, const orc_u256 *a, orc_u256 *r);
void orc_f_from_mont(int which, const orc_u256 *a, orc_u256 *r);
void orc_f_inverse(int which, const orc_u256 *a, orc_u256 *r);  /* Montgomery in/out */
void orc_f_constants(int which, orc_u256 *modulus, orc_u256 *R, orc_u256 *R2, uint64_t *np64);

/* ---- Fq2 ---- */
void orc_f2_mul(const orc_fq2 *a, const orc_fq2 *b, orc_fq2 *r);
void orc_f2_square(const orc_fq2 *a, orc_fq2 *r);
void orc_f2_inverse(const orc_fq2 *a, orc_fq2 *r);

/* ---- groups ---- */
void orc_g1_zero(orc_g1 *r);
int  orc_g1_is_zero(const orc_g1 *p);
int  orc_g1_eq(const orc_g1 *a, const orc_g1 *b);
void orc_g1_double(const orc_g1 *p, orc_g1 *r);
void orc_g1_add(const orc_g1 *a, const orc_g1 *b, orc_g1 *r);
void orc_g1_neg(const orc_g1 *p, orc_g1 *r);
void orc_g1_affine(const orc_g1 *p, orc_g1 *r);
void orc_g1_from_mont(const orc_g1 *p, orc_g1 *r);
void orc_g1_times_scalar(const orc_g1 *p, const uint8_t *scalar, int scalar_bytes, orc_g1 *r);

void orc_g2_zero(orc_g2 *r);
int  orc_g2_is_zero(const orc_g2 *p);
int  orc_g2_eq(const orc_g2 *a, const orc_g2 *b);
void orc_g2_double(const orc_g2 *p, orc_g2 *r);
void orc_g2_add(const orc_g2 *a, const orc_g2 *b, orc_g2 *r);
void orc_g2_neg(const orc_g2 *p, orc_g2 *r);
void orc_g2_affine(const orc_g2 *p, orc_g2 *r);
void orc_g2_from_mont(const orc_g2 *p, orc_g2 *r);
void orc_g2_times_scalar(const orc_g2 *p, const uint8_t *scalar, int scalar_bytes, orc_g2 *r);

/* ---- multiexp (reference algorithm, window w as in the reference: 7) ----
 * scalars: n x 32 B raw little-endian 256-bit (NOT reduced);
 * points : n x 64 B (G1) / 128 B (G2) affine Montgomery, x==0 => infinity.
 * The result is ACCUMULATED into *r (as the reference does). */
void orc_g1_multiexp2(const uint8_t *scalars, const uint8_t *points, uint32_t n, int w, orc_g1 *r);
void orc_g1_multiexp(const uint8_t *scalars, const uint8_t *points, uint32_t n, int w, orc_g1 *r);
void orc_g2_multiexp(const uint8_t *scalars, const uint8_t *points, uint32_t n, int w, orc_g2 *r);
/* host-level sharding of src/bn128.js:353-415 over `workers` threads */
void orc_g1_multiexp_workers(const uint8_t *scalars, const uint8_t *points, uint32_t n, int workers, orc_g1 *r);
void orc_g2_multiexp_workers(const uint8_t *scalars, const uint8_t *points, uint32_t n, int workers, orc_g2 *r);

/* ---- FFT over Fr (Montgomery, in place) ---- returns 0 ok, -1 if n is not
 * a power of two <= 2^28 (the reference traps there). */
int orc_fft(orc_u256 *x, uint32_t n, int odd);
int orc_ifft(orc_u256 *x, uint32_t n, int odd);
void orc_fr_to_mont_n(const orc_u256 *in, orc_u256 *out, uint32_t n);
void orc_fr_from_mont_n(const orc_u256 *in, orc_u256 *out, uint32_t n);

/* ---- pol_constructLC / CALC_H ---- */
/* returns number of bytes consumed from pols, or -1 on malformed input */
int64_t orc_pol_construct_lc(const uint8_t *pols, size_t pols_len, const orc_u256 *signals_mont,
                             uint32_t n_signals, orc_u256 *res, uint32_t domain);
/* signals plain; out_h = domain x 32 B plain */
int orc_calc_h(const uint8_t *signals, const uint8_t *polsA, size_t lenA, const uint8_t *polsB,
               size_t lenB, uint32_t n_signals, uint32_t domain, uint8_t *out_h);

/* ---- full prover ----
 * out: pi_a (x,y,z) 96 B | pi_b (x.c0,x.c1,y.c0,y.c1,z.c0,z.c1) 192 B |
 * pi_c 96 B, all affine and NOT Montgomery (what bin2g1/bin2g2 print).
 * r32/s32: the 32 raw random bytes each. workers = threads for the MSMs. */
int orc_groth16_prove(const uint8_t *witness, size_t witness_len, const uint8_t *pkey, size_t pkey_len,
                      const uint8_t *r32, const uint8_t *s32, int workers, uint8_t *out384);

#ifdef __cplusplus
}
#endif
#endif
