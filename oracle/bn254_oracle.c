/*
 * oracle/bn254_oracle.c — CPU restatement of the Spartan-BN254 prover hot path
 * (BN254 G1 MSM behind the Hyrax commitment + sumcheck round evaluations).
 *
 * TEST INFRASTRUCTURE ONLY — see bn254_oracle.h for who may use it and for the
 * parity status ("parity unpinned" at byte level by reference fixtures; pinned by
 * the reference's unit-test relations and an independent big-integer model).
 *
 * Written from the mathematical definitions and the reference's call sites
 * (cited per function as /root/reference/src/<file>:<lines>); the arkworks
 * internals it stands in for (ark-ff Fp256 Montgomery backend, ark-ec
 * short-Weierstrass Jacobian formulas, VariableBaseMSM) are third-party crates
 * pinned only as "0.5" (Cargo.toml:9-11) and absent from /root/reference.
 *
 * Representation here: 4 x 64-bit limbs, Montgomery form R = 2^256 (the device
 * code uses 8 x 32-bit limbs and different formulas, so the two are independent).
 */
#include "bn254_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;
typedef struct { fe p, r2, one; uint64_t ninv; } fctx;

/* SURVEY App. A constants (re-derived numerically; tests/test_oracle_golden.py::test_field_kat exercises them) */
static const fctx FQ = {
  {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}},
  {{0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full}},
  {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full}},
  0x87d20782e4866389ull};
static const fctx FR = {
  {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}},
  {{0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}},
  {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}},
  0xc2e1f593efffffffull};
static const fctx* ctx_of(int which) { return which ? &FR : &FQ; }

/* ------------------------------------------------------------------ field */
static int ge4(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
  return 1;
}
static uint64_t sub4(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
  return br;
}
static uint64_t add4(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t c = 0;
  for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + b[i] + c; o[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  return c;
}
static int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_eq(const fe* a, const fe* b) { return memcmp(a, b, sizeof(fe)) == 0; }
static void fe_add(const fctx* f, fe* o, const fe* a, const fe* b) {
  uint64_t c = add4(o->v, a->v, b->v);
  if (c || ge4(o->v, f->p.v)) sub4(o->v, o->v, f->p.v);
}
static void fe_sub(const fctx* f, fe* o, const fe* a, const fe* b) {
  if (sub4(o->v, a->v, b->v)) add4(o->v, o->v, f->p.v);
}
static void fe_neg(const fctx* f, fe* o, const fe* a) {
  if (fe_is_zero(a)) { *o = *a; return; }
  sub4(o->v, f->p.v, a->v);
}
static void fe_dbl(const fctx* f, fe* o, const fe* a) { fe_add(f, o, a, a); }
/* Montgomery product a*b*R^-1 mod p: schoolbook 512-bit product then word-wise reduction */
static void fe_mul(const fctx* f, fe* o, const fe* a, const fe* b) {
  uint64_t t[9] = {0};
  uint64_t w[8] = {0};
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 4; j++) { u128 s = (u128)a->v[i] * b->v[j] + w[i + j] + c; w[i + j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    w[i + 4] = c;
  }
  memcpy(t, w, sizeof w);
  for (int i = 0; i < 4; i++) {
    uint64_t m = t[i] * f->ninv, c = 0;
    for (int j = 0; j < 4; j++) { u128 s = (u128)m * f->p.v[j] + t[i + j] + c; t[i + j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    for (int k = i + 4; c && k < 9; k++) { u128 s = (u128)t[k] + c; t[k] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  }
  uint64_t r[4] = {t[4], t[5], t[6], t[7]};
  if (t[8] || ge4(r, f->p.v)) sub4(r, r, f->p.v);
  memcpy(o->v, r, sizeof r);
}
static void fe_sqr(const fctx* f, fe* o, const fe* a) { fe_mul(f, o, a, a); }
static void fe_to_mont(const fctx* f, fe* o, const fe* a) { fe_mul(f, o, a, &f->r2); }
static void fe_from_mont(const fctx* f, fe* o, const fe* a) { fe one = {{1, 0, 0, 0}}; fe_mul(f, o, a, &one); }
static void fe_pow(const fctx* f, fe* o, const fe* a, const uint64_t e[4]) {
  fe acc = f->one, base = *a;
  for (int i = 0; i < 256; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(f, &acc, &acc, &base);
    fe_sqr(f, &base, &base);
  }
  *o = acc;
}
static void fe_inv(const fctx* f, fe* o, const fe* a) { /* a^(p-2); 0 -> 0 */
  uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0}; sub4(e, f->p.v, two); fe_pow(f, o, a, e);
}
static void load_le(fe* o, const uint8_t b[32]) {
  for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int k = 7; k >= 0; k--) w = (w << 8) | b[8 * i + k]; o->v[i] = w; }
}
static void store_le(uint8_t b[32], const fe* a) {
  for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(a->v[i] >> (8 * k));
}
/* canonical bytes -> Montgomery (caller guarantees < modulus; non-canonical input is reduced once) */
static void fe_from_bytes(const fctx* f, fe* o, const uint8_t b[32]) {
  fe t; load_le(&t, b);
  while (ge4(t.v, f->p.v)) sub4(t.v, t.v, f->p.v);
  fe_to_mont(f, o, &t);
}
static void fe_to_bytes(const fctx* f, uint8_t b[32], const fe* a) { fe t; fe_from_mont(f, &t, a); store_le(b, &t); }
static void fe_from_u64(const fctx* f, fe* o, uint64_t x) { fe t = {{x, 0, 0, 0}}; fe_to_mont(f, o, &t); }

int orc_fe_is_canonical(int which, const uint8_t a[32]) { fe t; load_le(&t, a); return !ge4(t.v, ctx_of(which)->p.v); }
void orc_fe_add(int w, const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) { const fctx* f = ctx_of(w); fe x, y; fe_from_bytes(f, &x, a); fe_from_bytes(f, &y, b); fe_add(f, &x, &x, &y); fe_to_bytes(f, o, &x); }
void orc_fe_sub(int w, const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) { const fctx* f = ctx_of(w); fe x, y; fe_from_bytes(f, &x, a); fe_from_bytes(f, &y, b); fe_sub(f, &x, &x, &y); fe_to_bytes(f, o, &x); }
void orc_fe_mul(int w, const uint8_t a[32], const uint8_t b[32], uint8_t o[32]) { const fctx* f = ctx_of(w); fe x, y; fe_from_bytes(f, &x, a); fe_from_bytes(f, &y, b); fe_mul(f, &x, &x, &y); fe_to_bytes(f, o, &x); }
void orc_fe_inv(int w, const uint8_t a[32], uint8_t o[32]) { const fctx* f = ctx_of(w); fe x; fe_from_bytes(f, &x, a); fe_inv(f, &x, &x); fe_to_bytes(f, o, &x); }
void orc_fr_from_wide(const uint8_t in[64], uint8_t out[32]) {
  /* value = lo + hi*2^256 ; in Montgomery arithmetic: mont(lo)*1 + mont(hi)*mont(2^256)   (transcript.rs:56-67) */
  fe lo, hi, t; fe_from_bytes(&FR, &lo, in); fe_from_bytes(&FR, &hi, in + 32);
  fe_mul(&FR, &t, &hi, &FR.r2); /* hi_m * R2 * R^-1 = hi*R * R = mont(hi * 2^256) */
  fe_add(&FR, &lo, &lo, &t); fe_to_bytes(&FR, out, &lo);
}
void orc_fr_dot(const uint8_t* k, const uint8_t* s, size_t n, uint8_t out[32]) {
  fe acc = {{0, 0, 0, 0}};
  for (size_t i = 0; i < n; i++) { fe a, b; fe_from_bytes(&FR, &a, k + 32 * i); fe_from_bytes(&FR, &b, s + 32 * i); fe_mul(&FR, &a, &a, &b); fe_add(&FR, &acc, &acc, &a); }
  fe_to_bytes(&FR, out, &acc);
}
void orc_dotproduct(const uint8_t* a, const uint8_t* b, size_t n, uint8_t out[32]) { orc_fr_dot(a, b, n, out); } /* hyrax.rs:409-415 */

/* ------------------------------------------------------------------ G1: y^2 = x^3 + 3, Jacobian */
typedef struct { fe X, Y, Z; } jac;   /* Z == 0 <=> infinity */
typedef struct { fe x, y; int inf; } aff;
static const fe* Q(void) { return &FQ.p; }
static void jac_set_inf(jac* p) { memset(p, 0, sizeof *p); p->X = FQ.one; p->Y = FQ.one; }
static int jac_is_inf(const jac* p) { return fe_is_zero(&p->Z); }
static void jac_from_aff(jac* o, const aff* a) { if (a->inf) { jac_set_inf(o); return; } o->X = a->x; o->Y = a->y; o->Z = FQ.one; }
static void jac_double(jac* o, const jac* p) {
  if (jac_is_inf(p) || fe_is_zero(&p->Y)) { jac_set_inf(o); return; }
  const fctx* f = &FQ; fe A, B, C, D, E, F, t, X3, Y3, Z3;
  fe_sqr(f, &A, &p->X); fe_sqr(f, &B, &p->Y); fe_sqr(f, &C, &B);
  fe_add(f, &t, &p->X, &B); fe_sqr(f, &t, &t); fe_sub(f, &t, &t, &A); fe_sub(f, &t, &t, &C); fe_dbl(f, &D, &t);
  fe_dbl(f, &E, &A); fe_add(f, &E, &E, &A); fe_sqr(f, &F, &E);
  fe_dbl(f, &t, &D); fe_sub(f, &X3, &F, &t);
  fe_sub(f, &t, &D, &X3); fe_mul(f, &Y3, &E, &t); fe_dbl(f, &t, &C); fe_dbl(f, &t, &t); fe_dbl(f, &t, &t); fe_sub(f, &Y3, &Y3, &t);
  fe_mul(f, &Z3, &p->Y, &p->Z); fe_dbl(f, &Z3, &Z3);
  o->X = X3; o->Y = Y3; o->Z = Z3; (void)Q;
}
static void jac_add(jac* o, const jac* p, const jac* q) {
  if (jac_is_inf(p)) { *o = *q; return; }
  if (jac_is_inf(q)) { *o = *p; return; }
  const fctx* f = &FQ; fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, R, t, HH, HHH, V, X3, Y3, Z3;
  fe_sqr(f, &Z1Z1, &p->Z); fe_sqr(f, &Z2Z2, &q->Z);
  fe_mul(f, &U1, &p->X, &Z2Z2); fe_mul(f, &U2, &q->X, &Z1Z1);
  fe_mul(f, &t, &q->Z, &Z2Z2); fe_mul(f, &S1, &p->Y, &t);
  fe_mul(f, &t, &p->Z, &Z1Z1); fe_mul(f, &S2, &q->Y, &t);
  if (fe_eq(&U1, &U2)) { if (fe_eq(&S1, &S2)) { jac_double(o, p); } else { jac_set_inf(o); } return; }
  fe_sub(f, &H, &U2, &U1); fe_sub(f, &R, &S2, &S1);
  fe_sqr(f, &HH, &H); fe_mul(f, &HHH, &HH, &H); fe_mul(f, &V, &U1, &HH);
  fe_sqr(f, &X3, &R); fe_sub(f, &X3, &X3, &HHH); fe_sub(f, &X3, &X3, &V); fe_sub(f, &X3, &X3, &V);
  fe_sub(f, &t, &V, &X3); fe_mul(f, &Y3, &R, &t); fe_mul(f, &t, &S1, &HHH); fe_sub(f, &Y3, &Y3, &t);
  fe_mul(f, &Z3, &p->Z, &q->Z); fe_mul(f, &Z3, &Z3, &H);
  o->X = X3; o->Y = Y3; o->Z = Z3;
}
static void jac_add_aff(jac* o, const jac* p, const aff* q) {
  if (q->inf) { *o = *p; return; }
  if (jac_is_inf(p)) { jac_from_aff(o, q); return; }
  const fctx* f = &FQ; fe Z1Z1, U2, S2, H, R, t, HH, HHH, V, X3, Y3, Z3;
  fe_sqr(f, &Z1Z1, &p->Z); fe_mul(f, &U2, &q->x, &Z1Z1);
  fe_mul(f, &t, &p->Z, &Z1Z1); fe_mul(f, &S2, &q->y, &t);
  if (fe_eq(&p->X, &U2)) { if (fe_eq(&p->Y, &S2)) { jac_double(o, p); } else { jac_set_inf(o); } return; }
  fe_sub(f, &H, &U2, &p->X); fe_sub(f, &R, &S2, &p->Y);
  fe_sqr(f, &HH, &H); fe_mul(f, &HHH, &HH, &H); fe_mul(f, &V, &p->X, &HH);
  fe_sqr(f, &X3, &R); fe_sub(f, &X3, &X3, &HHH); fe_sub(f, &X3, &X3, &V); fe_sub(f, &X3, &X3, &V);
  fe_sub(f, &t, &V, &X3); fe_mul(f, &Y3, &R, &t); fe_mul(f, &t, &p->Y, &HHH); fe_sub(f, &Y3, &Y3, &t);
  fe_mul(f, &Z3, &p->Z, &H);
  o->X = X3; o->Y = Y3; o->Z = Z3;
}
static void aff_neg(aff* o, const aff* a) { *o = *a; if (!a->inf) fe_neg(&FQ, &o->y, &a->y); }
static void jac_to_aff(aff* o, const jac* p) {
  if (jac_is_inf(p)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
  const fctx* f = &FQ; fe zi, zi2, zi3;
  fe_inv(f, &zi, &p->Z); fe_sqr(f, &zi2, &zi); fe_mul(f, &zi3, &zi2, &zi);
  fe_mul(f, &o->x, &p->X, &zi2); fe_mul(f, &o->y, &p->Y, &zi3); o->inf = 0;
}
static void aff_from_bytes(aff* o, const uint8_t b[64]) {
  int z = 1; for (int i = 0; i < 64; i++) if (b[i]) { z = 0; break; }
  if (z) { memset(o, 0, sizeof *o); o->inf = 1; return; }
  fe_from_bytes(&FQ, &o->x, b); fe_from_bytes(&FQ, &o->y, b + 32); o->inf = 0;
}
static void aff_to_bytes(uint8_t b[64], const aff* a) {
  if (a->inf) { memset(b, 0, 64); return; }
  fe_to_bytes(&FQ, b, &a->x); fe_to_bytes(&FQ, b + 32, &a->y);
}
static void jac_to_bytes(uint8_t b[64], const jac* p) { aff a; jac_to_aff(&a, p); aff_to_bytes(b, &a); }
/* k given as canonical integer limbs (not Montgomery) */
static void jac_mul_int(jac* o, const jac* p, const uint64_t k[4]) {
  jac acc; jac_set_inf(&acc);
  for (int i = 255; i >= 0; i--) { jac_double(&acc, &acc); if ((k[i >> 6] >> (i & 63)) & 1) jac_add(&acc, &acc, p); }
  *o = acc;
}
static void gen_aff(aff* g) { fe_from_u64(&FQ, &g->x, 1); fe_from_u64(&FQ, &g->y, 2); g->inf = 0; }

int orc_g1_on_curve(const uint8_t p[64]) {
  aff a; aff_from_bytes(&a, p); if (a.inf) return 1;
  if (!orc_fe_is_canonical(0, p) || !orc_fe_is_canonical(0, p + 32)) return 0;
  fe l, r, three; fe_sqr(&FQ, &l, &a.y); fe_sqr(&FQ, &r, &a.x); fe_mul(&FQ, &r, &r, &a.x); fe_from_u64(&FQ, &three, 3); fe_add(&FQ, &r, &r, &three);
  return fe_eq(&l, &r);
}
void orc_g1_add(const uint8_t p[64], const uint8_t q[64], uint8_t out[64]) { aff a, b; jac j; aff_from_bytes(&a, p); aff_from_bytes(&b, q); jac_from_aff(&j, &a); jac_add_aff(&j, &j, &b); jac_to_bytes(out, &j); }
void orc_g1_neg(const uint8_t p[64], uint8_t out[64]) { aff a; aff_from_bytes(&a, p); aff_neg(&a, &a); aff_to_bytes(out, &a); }
void orc_g1_mul(const uint8_t p[64], const uint8_t k[32], uint8_t out[64]) { aff a; jac j; fe s; aff_from_bytes(&a, p); jac_from_aff(&j, &a); load_le(&s, k); jac_mul_int(&j, &j, s.v); jac_to_bytes(out, &j); }
void orc_g1_compress(const uint8_t p[64], uint8_t out[32]) {
  aff a; aff_from_bytes(&a, p);
  if (a.inf) { memset(out, 0, 32); out[31] = 0x40; return; }
  memcpy(out, p, 32);
  fe y, ny; load_le(&y, p + 32); sub4(ny.v, FQ.p.v, y.v);   /* ny = p - y (canonical ints) */
  if (!ge4(ny.v, y.v)) out[31] |= 0x80;                     /* y > p - y */
}
int orc_g1_decompress(const uint8_t in[32], uint8_t out[64]) {
  if (in[31] & 0x40) { memset(out, 0, 64); return 1; }
  uint8_t xb[32]; memcpy(xb, in, 32); xb[31] &= 0x3f;
  if (!orc_fe_is_canonical(0, xb)) return 0;
  fe x, rhs, three, y, chk; fe_from_bytes(&FQ, &x, xb);
  fe_sqr(&FQ, &rhs, &x); fe_mul(&FQ, &rhs, &rhs, &x); fe_from_u64(&FQ, &three, 3); fe_add(&FQ, &rhs, &rhs, &three);
  /* p = 3 mod 4 : sqrt = rhs^((p+1)/4) */
  uint64_t e[4]; uint64_t one[4] = {1, 0, 0, 0}; add4(e, FQ.p.v, one);
  for (int i = 0; i < 4; i++) e[i] = (e[i] >> 2) | (i < 3 ? e[i + 1] << 62 : 0);
  fe_pow(&FQ, &y, &rhs, e); fe_sqr(&FQ, &chk, &y);
  if (!fe_eq(&chk, &rhs)) return 0;
  uint8_t yb[32]; fe_to_bytes(&FQ, yb, &y);
  fe yi, nyi; load_le(&yi, yb); sub4(nyi.v, FQ.p.v, yi.v);
  int is_neg = !ge4(nyi.v, yi.v); int want_neg = (in[31] & 0x80) != 0;
  if (is_neg != want_neg) store_le(yb, &nyi);
  memcpy(out, xb, 32); memcpy(out + 32, yb, 32); return 1;
}
void orc_g1_mul_gen_batch(const uint8_t* s, size_t n, uint8_t* out_xy, int threads) {
  aff g; gen_aff(&g); jac G; jac_from_aff(&G, &g);
  (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 16)
  for (long i = 0; i < (long)n; i++) { fe k; load_le(&k, s + 32 * i); jac r; jac_mul_int(&r, &G, k.v); jac_to_bytes(out_xy + 64 * i, &r); }
}

/* ------------------------------------------------------------------ MSM */
void orc_msm_naive(const uint8_t* scalars, const uint8_t* bases, size_t n, uint8_t out[64]) {
  jac acc; jac_set_inf(&acc);
  for (size_t i = 0; i < n; i++) {
    aff a; aff_from_bytes(&a, bases + 64 * i); jac p, t; jac_from_aff(&p, &a);
    fe k; load_le(&k, scalars + 32 * i); jac_mul_int(&t, &p, k.v); jac_add(&acc, &acc, &t);
  }
  jac_to_bytes(out, &acc);
}
/* ark-ec 0.5 window heuristic: c = 3 if n < 32 else floor(ceil(log2 n) * 69 / 100) + 2 */
int orc_msm_window_bits(size_t n) {
  if (n < 32) return 3;
  int lg = 0; while (((size_t)1 << lg) < n) lg++;
  return lg * 69 / 100 + 2;
}
/* signed radix-2^c digits of a canonical 254-bit integer, digit in [-2^(c-1), 2^(c-1)), last digit absorbs the carry */
static void make_digits(const uint64_t k[4], int c, int ndig, int64_t* out) {
  uint64_t carry = 0, radix = 1ull << c, mask = radix - 1;
  for (int i = 0; i < ndig; i++) {
    int bit = i * c, w = bit >> 6, sh = bit & 63;
    uint64_t buf = w < 4 ? k[w] >> sh : 0;
    if (sh + c > 64 && w + 1 < 4) buf |= k[w + 1] << (64 - sh);
    uint64_t coef = carry + (buf & mask);
    carry = (coef + radix / 2) >> c;
    int64_t d = (int64_t)coef - (int64_t)(carry << c);
    if (i == ndig - 1) d += (int64_t)(carry << c);
    out[i] = d;
  }
}
static void msm_core(const fe* k, const aff* P, size_t n, jac* result, int threads) {
  if (n == 0) { jac_set_inf(result); return; }
  int c = orc_msm_window_bits(n), ndig = (254 + c - 1) / c;
  size_t nb = (size_t)1 << (c - 1);
  int64_t* dig = (int64_t*)malloc(sizeof(int64_t) * n * ndig);
  for (size_t i = 0; i < n; i++) make_digits(k[i].v, c, ndig, dig + i * ndig);
  jac* wsum = (jac*)malloc(sizeof(jac) * ndig);
  (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
  for (int w = 0; w < ndig; w++) {
    jac* b = (jac*)malloc(sizeof(jac) * nb);
    for (size_t j = 0; j < nb; j++) jac_set_inf(&b[j]);
    for (size_t i = 0; i < n; i++) {
      int64_t d = dig[i * ndig + w];
      if (d > 0) jac_add_aff(&b[d - 1], &b[d - 1], &P[i]);
      else if (d < 0) { aff m; aff_neg(&m, &P[i]); jac_add_aff(&b[-d - 1], &b[-d - 1], &m); }
    }
    jac run, res; jac_set_inf(&run); jac_set_inf(&res);
    for (size_t j = nb; j-- > 0;) { jac_add(&run, &run, &b[j]); jac_add(&res, &res, &run); }
    wsum[w] = res; free(b);
  }
  jac total; jac_set_inf(&total);
  for (int w = ndig - 1; w >= 1; w--) { jac_add(&total, &total, &wsum[w]); for (int d = 0; d < c; d++) jac_double(&total, &total); }
  jac_add(&total, &total, &wsum[0]);
  *result = total; free(wsum); free(dig);
}
void orc_msm_pippenger(const uint8_t* scalars, const uint8_t* bases, size_t n, uint8_t out[64], int threads) {
  fe* k = (fe*)malloc(sizeof(fe) * (n ? n : 1)); aff* P = (aff*)malloc(sizeof(aff) * (n ? n : 1));
  for (size_t i = 0; i < n; i++) { load_le(&k[i], scalars + 32 * i); aff_from_bytes(&P[i], bases + 64 * i); }
  jac r; msm_core(k, P, n, &r, threads); jac_to_bytes(out, &r); free(k); free(P);
}

/* ------------------------------------------------------------------ Keccak-f[1600] (FIPS 202) */
static const uint64_t KRC[24] = {
  0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
  0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
  0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
  0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
static uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
static void keccakf(uint64_t s[25]) {
  for (int r = 0; r < 24; r++) {
    uint64_t bc[5], t;
    for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
    for (int i = 0; i < 5; i++) { t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1); for (int j = 0; j < 25; j += 5) s[j + i] ^= t; }
    t = s[1];
    for (int i = 0; i < 24; i++) { int j = KPIL[i]; uint64_t b = s[j]; s[j] = rotl64(t, KROT[i]); t = b; }
    for (int j = 0; j < 25; j += 5) { for (int i = 0; i < 5; i++) bc[i] = s[j + i]; for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5]; }
    s[0] ^= KRC[r];
  }
}
typedef struct { uint64_t s[25]; uint8_t buf[200]; size_t rate, pos; int squeezing; uint8_t suffix; } sponge;
static void sp_init(sponge* k, size_t rate, uint8_t suffix) { memset(k, 0, sizeof *k); k->rate = rate; k->suffix = suffix; }
static void sp_xor_block(sponge* k) { for (size_t i = 0; i < k->rate / 8; i++) { uint64_t w = 0; for (int b = 7; b >= 0; b--) w = (w << 8) | k->buf[8 * i + b]; k->s[i] ^= w; } }
static void sp_absorb(sponge* k, const uint8_t* in, size_t len) {
  for (size_t i = 0; i < len; i++) { k->buf[k->pos++] = in[i]; if (k->pos == k->rate) { sp_xor_block(k); keccakf(k->s); k->pos = 0; memset(k->buf, 0, 200); } }
}
static void sp_squeeze(sponge* k, uint8_t* out, size_t len) {
  if (!k->squeezing) {
    memset(k->buf + k->pos, 0, k->rate - k->pos); k->buf[k->pos] ^= k->suffix; k->buf[k->rate - 1] ^= 0x80;
    sp_xor_block(k); keccakf(k->s); k->pos = 0; k->squeezing = 1;
  }
  for (size_t i = 0; i < len; i++) {
    if (k->pos == k->rate) { keccakf(k->s); k->pos = 0; }
    out[i] = (uint8_t)(k->s[k->pos / 8] >> (8 * (k->pos % 8))); k->pos++;
  }
}
void orc_sha3_256(const uint8_t* in, size_t len, uint8_t out[32]) { sponge k; sp_init(&k, 136, 0x06); sp_absorb(&k, in, len); sp_squeeze(&k, out, 32); }
void orc_shake256(const uint8_t* in, size_t len, uint8_t* out, size_t outlen) { sponge k; sp_init(&k, 136, 0x1f); sp_absorb(&k, in, len); sp_squeeze(&k, out, outlen); }

/* ------------------------------------------------------------------ generators + commit */
/* group.rs:110-131: s = LE(SHA3-256(bytes)); if s >= r: s = LE(SHA3-256("fallback"||bytes)); if still >= r: s = 1 */
static void uniform_bytes_to_dlog(const uint8_t ub[64], uint8_t s[32]) {
  orc_sha3_256(ub, 64, s);
  if (orc_fe_is_canonical(1, s)) return;
  uint8_t tmp[72]; memcpy(tmp, "fallback", 8); memcpy(tmp + 8, ub, 64);
  orc_sha3_256(tmp, 72, s);
  if (orc_fe_is_canonical(1, s)) return;
  memset(s, 0, 32); s[0] = 1;
}
/* commitments.rs:31-62: SHAKE256(label || compressed basepoint) stream, 64 B per generator, n+1 generators, h last */
void orc_gens_new(size_t n, const uint8_t* label, size_t label_len, uint8_t* out_xy, uint8_t* out_dlogs, int threads) {
  uint8_t gxy[64] = {0}, gc[32]; gxy[0] = 1; gxy[32] = 2; orc_g1_compress(gxy, gc);
  sponge k; sp_init(&k, 136, 0x1f); sp_absorb(&k, label, label_len); sp_absorb(&k, gc, 32);
  uint8_t* dl = (uint8_t*)malloc(32 * (n + 1));
  for (size_t i = 0; i <= n; i++) { uint8_t ub[64]; sp_squeeze(&k, ub, 64); uniform_bytes_to_dlog(ub, dl + 32 * i); }
  orc_g1_mul_gen_batch(dl, n + 1, out_xy, threads);
  if (out_dlogs) memcpy(out_dlogs, dl, 32 * (n + 1));
  free(dl);
}
/* commitments.rs:144-154: scalars.push(blind); points.push(h); msm_affine */
static void commit_core(const uint8_t* scalars, size_t n, const uint8_t* blind, const aff* GH /* n+1 */, jac* out) {
  fe* k = (fe*)malloc(sizeof(fe) * (n + 1));
  for (size_t i = 0; i < n; i++) load_le(&k[i], scalars + 32 * i);
  if (blind) load_le(&k[n], blind); else memset(&k[n], 0, sizeof(fe));
  msm_core(k, GH, n + 1, out, 1); free(k);
}
void orc_commit(const uint8_t* scalars, size_t n, const uint8_t blind[32], const uint8_t* G_xy, const uint8_t h_xy[64], uint8_t out[64]) {
  aff* GH = (aff*)malloc(sizeof(aff) * (n + 1));
  for (size_t i = 0; i < n; i++) aff_from_bytes(&GH[i], G_xy + 64 * i);
  aff_from_bytes(&GH[n], h_xy);
  jac r; commit_core(scalars, n, blind, GH, &r); jac_to_bytes(out, &r); free(GH);
}
/* hyrax.rs:253-267 (rayon over rows) */
void orc_commit_rows(const uint8_t* Z, const uint8_t* blinds, size_t L, size_t R, const uint8_t* G_xy, const uint8_t h_xy[64], uint8_t* out_xy, int threads) {
  aff* GH = (aff*)malloc(sizeof(aff) * (R + 1));
  for (size_t i = 0; i < R; i++) aff_from_bytes(&GH[i], G_xy + 64 * i);
  aff_from_bytes(&GH[R], h_xy);
  (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
  for (long i = 0; i < (long)L; i++) { jac r; commit_core(Z + 32 * R * i, R, blinds ? blinds + 32 * i : NULL, GH, &r); jac_to_bytes(out_xy + 64 * i, &r); }
  free(GH);
}
void orc_factored_lens(size_t ell, size_t* left, size_t* right) { *left = ell / 2; *right = ell - ell / 2; }

/* ------------------------------------------------------------------ multilinear / sumcheck (Fr) */
/* hyrax.rs:355-369 */
void orc_eq_evals(const uint8_t* r, size_t ell, uint8_t* out) {
  size_t N = (size_t)1 << ell; fe* ev = (fe*)malloc(sizeof(fe) * N);
  for (size_t i = 0; i < N; i++) ev[i] = FR.one;
  size_t size = 1;
  for (size_t j = 0; j < ell; j++) {
    fe rj; fe_from_bytes(&FR, &rj, r + 32 * j); size *= 2;
    for (size_t i = size; i-- > 0;) { if (!(i & 1)) continue; fe s = ev[i / 2]; fe_mul(&FR, &ev[i], &s, &rj); fe_sub(&FR, &ev[i - 1], &s, &ev[i]); }
  }
  for (size_t i = 0; i < N; i++) fe_to_bytes(&FR, out + 32 * i, &ev[i]);
  free(ev);
}
/* sparse_mlpoly_full.rs:756-758: hash_func(addr, val, ts) = ts*r_hash^2 + val*r_hash + addr, minus r_multiset_check */
void orc_hash_layer(const uint32_t* addr, const uint8_t* val, const uint32_t* ts, uint32_t ts_add, const uint8_t g[32], const uint8_t tau[32], size_t n, uint8_t* out) {
  fe gg, g2, tt; fe_from_bytes(&FR, &gg, g); fe_sqr(&FR, &g2, &gg); fe_from_bytes(&FR, &tt, tau);
  for (size_t j = 0; j < n; j++) {
    fe a, v, t, acc;
    fe_from_u64(&FR, &a, addr ? addr[j] : (uint64_t)j); fe_from_bytes(&FR, &v, val + 32 * j); fe_from_u64(&FR, &t, (uint64_t)(ts ? ts[j] : 0) + ts_add);
    fe_mul(&FR, &acc, &t, &g2); fe_mul(&FR, &v, &v, &gg); fe_add(&FR, &acc, &acc, &v); fe_add(&FR, &acc, &acc, &a); fe_sub(&FR, &acc, &acc, &tt);
    fe_to_bytes(&FR, out + 32 * j, &acc);
  }
}
/* product_tree.rs:21-37 */
void orc_product_layer(const uint8_t* in, size_t len, uint8_t* out) {
  for (size_t i = 0; i < len / 2; i++) { fe a, b; fe_from_bytes(&FR, &a, in + 32 * i); fe_from_bytes(&FR, &b, in + 32 * (i + len / 2)); fe_mul(&FR, &a, &a, &b); fe_to_bytes(&FR, out + 32 * i, &a); }
}
/* hyrax.rs:311-324 */
void orc_bound(const uint8_t* Z, const uint8_t* L, size_t L_size, size_t R_size, uint8_t* out) {
  for (size_t i = 0; i < R_size; i++) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t j = 0; j < L_size; j++) { fe l, z; fe_from_bytes(&FR, &l, L + 32 * j); fe_from_bytes(&FR, &z, Z + 32 * (j * R_size + i)); fe_mul(&FR, &z, &z, &l); fe_add(&FR, &acc, &acc, &z); }
    fe_to_bytes(&FR, out + 32 * i, &acc);
  }
}
/* hyrax.rs:195-203 */
void orc_bind_top(uint8_t* Z, size_t len, const uint8_t r[32]) {
  size_t n = len / 2; fe rr; fe_from_bytes(&FR, &rr, r);
  for (size_t i = 0; i < n; i++) { fe lo, hi, t; fe_from_bytes(&FR, &lo, Z + 32 * i); fe_from_bytes(&FR, &hi, Z + 32 * (i + n)); fe_sub(&FR, &t, &hi, &lo); fe_mul(&FR, &t, &t, &rr); fe_add(&FR, &lo, &lo, &t); fe_to_bytes(&FR, Z + 32 * i, &lo); }
}
static void tab_load(fe* lo, fe* hi, const uint8_t* T, size_t i, size_t half) { fe_from_bytes(&FR, lo, T + 32 * i); fe_from_bytes(&FR, hi, T + 32 * (i + half)); }
/* p2 = 2*hi - lo ; p3 = p2 + hi - lo  (sumcheck.rs:116-126) */
static void pts(fe* p2, fe* p3, const fe* lo, const fe* hi) { fe_add(&FR, p2, hi, hi); fe_sub(&FR, p2, p2, lo); fe_add(&FR, p3, p2, hi); fe_sub(&FR, p3, p3, lo); }
void orc_sc_eval_cubic(const uint8_t* A, const uint8_t* B, const uint8_t* C, size_t len, uint8_t out[96]) {
  size_t h = len / 2; fe e0 = {{0}}, e2 = {{0}}, e3 = {{0}};
  for (size_t i = 0; i < h; i++) {
    fe al, ah, bl, bh, cl, ch, a2, a3, b2, b3, c2, c3, t;
    tab_load(&al, &ah, A, i, h); tab_load(&bl, &bh, B, i, h); tab_load(&cl, &ch, C, i, h);
    fe_mul(&FR, &t, &al, &bl); fe_mul(&FR, &t, &t, &cl); fe_add(&FR, &e0, &e0, &t);
    pts(&a2, &a3, &al, &ah); pts(&b2, &b3, &bl, &bh); pts(&c2, &c3, &cl, &ch);
    fe_mul(&FR, &t, &a2, &b2); fe_mul(&FR, &t, &t, &c2); fe_add(&FR, &e2, &e2, &t);
    fe_mul(&FR, &t, &a3, &b3); fe_mul(&FR, &t, &t, &c3); fe_add(&FR, &e3, &e3, &t);
  }
  fe_to_bytes(&FR, out, &e0); fe_to_bytes(&FR, out + 32, &e2); fe_to_bytes(&FR, out + 64, &e3);
}
static void comb_r1cs(fe* o, const fe* t, const fe* a, const fe* b, const fe* c) { fe x; fe_mul(&FR, &x, a, b); fe_sub(&FR, &x, &x, c); fe_mul(&FR, o, t, &x); } /* r1csproof.rs:288-292 */
void orc_sc_eval_r1cs(const uint8_t* T, const uint8_t* A, const uint8_t* B, const uint8_t* C, size_t len, uint8_t out[96]) {
  size_t h = len / 2; fe e0 = {{0}}, e2 = {{0}}, e3 = {{0}};
  for (size_t i = 0; i < h; i++) {
    fe tl, th, al, ah, bl, bh, cl, ch, t2, t3, a2, a3, b2, b3, c2, c3, x;
    tab_load(&tl, &th, T, i, h); tab_load(&al, &ah, A, i, h); tab_load(&bl, &bh, B, i, h); tab_load(&cl, &ch, C, i, h);
    comb_r1cs(&x, &tl, &al, &bl, &cl); fe_add(&FR, &e0, &e0, &x);
    pts(&t2, &t3, &tl, &th); pts(&a2, &a3, &al, &ah); pts(&b2, &b3, &bl, &bh); pts(&c2, &c3, &cl, &ch);
    comb_r1cs(&x, &t2, &a2, &b2, &c2); fe_add(&FR, &e2, &e2, &x);
    comb_r1cs(&x, &t3, &a3, &b3, &c3); fe_add(&FR, &e3, &e3, &x);
  }
  fe_to_bytes(&FR, out, &e0); fe_to_bytes(&FR, out + 32, &e2); fe_to_bytes(&FR, out + 64, &e3);
}
void orc_sc_eval_quad(const uint8_t* Z, const uint8_t* ABC, size_t len, uint8_t out[64]) {
  size_t h = len / 2; fe e0 = {{0}}, e2 = {{0}};
  for (size_t i = 0; i < h; i++) {
    fe zl, zh, al, ah, z2, a2, t;
    tab_load(&zl, &zh, Z, i, h); tab_load(&al, &ah, ABC, i, h);
    fe_mul(&FR, &t, &zl, &al); fe_add(&FR, &e0, &e0, &t);
    fe_add(&FR, &z2, &zh, &zh); fe_sub(&FR, &z2, &z2, &zl); fe_add(&FR, &a2, &ah, &ah); fe_sub(&FR, &a2, &a2, &al);
    fe_mul(&FR, &t, &z2, &a2); fe_add(&FR, &e2, &e2, &t);
  }
  fe_to_bytes(&FR, out, &e0); fe_to_bytes(&FR, out + 32, &e2);
}
/* SumcheckInstanceProof::prove_cubic_batched (sumcheck.rs:165-330) with comb_func = A*B*C (every call site: product_tree.rs:275-278)
 * and the Fiat-Shamir challenges r_j supplied by the caller (the reference squeezes them from the transcript, sumcheck.rs:282-286;
 * the transcript is outside the path).  Per round, in the reference's order: the "par" instances against the shared C
 * (:201-235), the "seq" instances with their own C (:238-267), the coeffs-weighted combination (:269-271), then every table bound
 * to r_j (:289-299).  Outputs: evals = rounds x (n_par + n_seq) x (e0, e2, e3); combined = rounds x (e0, e2, e3) weighted;
 * finals = A_par[i][0].., B_par[i][0].., C_par[0], A_seq.., B_seq.., C_seq.. (:302-318).  Field addition is associative, so the
 * threaded partial sums give the same canonical values as the reference's left-to-right loop. */
static void tab_to_fe(fe* o, const uint8_t* b, size_t len, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (size_t i = 0; i < len; i++) fe_from_bytes(&FR, &o[i], b + 32 * i);
}
static void cubic_sums(fe out[3], const fe* A, const fe* B, const fe* C, size_t h, int threads) {
  fe s0 = {{0}}, s2 = {{0}}, s3 = {{0}};
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    fe e0 = {{0}}, e2 = {{0}}, e3 = {{0}};
#pragma omp for schedule(static) nowait
    for (size_t i = 0; i < h; i++) {
      fe a2, a3, b2, b3, c2, c3, t;
      fe_mul(&FR, &t, &A[i], &B[i]); fe_mul(&FR, &t, &t, &C[i]); fe_add(&FR, &e0, &e0, &t);
      pts(&a2, &a3, &A[i], &A[i + h]); pts(&b2, &b3, &B[i], &B[i + h]); pts(&c2, &c3, &C[i], &C[i + h]);
      fe_mul(&FR, &t, &a2, &b2); fe_mul(&FR, &t, &t, &c2); fe_add(&FR, &e2, &e2, &t);
      fe_mul(&FR, &t, &a3, &b3); fe_mul(&FR, &t, &t, &c3); fe_add(&FR, &e3, &e3, &t);
    }
#pragma omp critical
    { fe_add(&FR, &s0, &s0, &e0); fe_add(&FR, &s2, &s2, &e2); fe_add(&FR, &s3, &s3, &e3); }
  }
  out[0] = s0; out[1] = s2; out[2] = s3;
}
static void bind_fe(fe* Z, size_t h, const fe* r, int threads) {          /* hyrax.rs:195-203 */
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (size_t i = 0; i < h; i++) { fe t; fe_sub(&FR, &t, &Z[i + h], &Z[i]); fe_mul(&FR, &t, &t, r); fe_add(&FR, &Z[i], &Z[i], &t); }
}
void orc_sc_prove_cubic_batched(const uint8_t* const* A_par, const uint8_t* const* B_par, const uint8_t* C_par, size_t n_par,
                                const uint8_t* const* A_seq, const uint8_t* const* B_seq, const uint8_t* const* C_seq, size_t n_seq,
                                size_t len, size_t num_rounds, const uint8_t* coeffs, const uint8_t* challenges,
                                uint8_t* evals, uint8_t* combined, uint8_t* finals, int threads) {
  const size_t ninst = n_par + n_seq, ntab = 2 * n_par + (n_par ? 1 : 0) + 3 * n_seq;
  fe** T = (fe**)malloc(sizeof(fe*) * (ntab ? ntab : 1));
  /* table order = the order of `finals` */
  size_t k = 0;
  for (size_t i = 0; i < n_par; i++) { T[k] = (fe*)malloc(sizeof(fe) * len); tab_to_fe(T[k++], A_par[i], len, threads); }
  for (size_t i = 0; i < n_par; i++) { T[k] = (fe*)malloc(sizeof(fe) * len); tab_to_fe(T[k++], B_par[i], len, threads); }
  if (n_par) { T[k] = (fe*)malloc(sizeof(fe) * len); tab_to_fe(T[k++], C_par, len, threads); }
  for (size_t i = 0; i < n_seq; i++) { T[k] = (fe*)malloc(sizeof(fe) * len); tab_to_fe(T[k++], A_seq[i], len, threads); }
  for (size_t i = 0; i < n_seq; i++) { T[k] = (fe*)malloc(sizeof(fe) * len); tab_to_fe(T[k++], B_seq[i], len, threads); }
  for (size_t i = 0; i < n_seq; i++) { T[k] = (fe*)malloc(sizeof(fe) * len); tab_to_fe(T[k++], C_seq[i], len, threads); }
  fe* co = (fe*)malloc(sizeof(fe) * (ninst ? ninst : 1));
  for (size_t i = 0; i < ninst; i++) fe_from_bytes(&FR, &co[i], coeffs + 32 * i);
  const size_t oCp = 2 * n_par, oAs = oCp + (n_par ? 1 : 0), oBs = oAs + n_seq, oCs = oBs + n_seq;
  size_t cur = len;
  for (size_t j = 0; j < num_rounds && cur >= 2; j++) {
    const size_t h = cur / 2;
    fe comb[3] = {{{0}}, {{0}}, {{0}}};
    for (size_t i = 0; i < ninst; i++) {
      fe e[3];
      if (i < n_par) cubic_sums(e, T[i], T[n_par + i], T[oCp], h, threads);
      else { const size_t q = i - n_par; cubic_sums(e, T[oAs + q], T[oBs + q], T[oCs + q], h, threads); }
      for (int t = 0; t < 3; t++) {
        if (evals) fe_to_bytes(&FR, evals + 96 * (j * ninst + i) + 32 * t, &e[t]);
        fe w; fe_mul(&FR, &w, &e[t], &co[i]); fe_add(&FR, &comb[t], &comb[t], &w);          /* sumcheck.rs:269-271 */
      }
    }
    if (combined) for (int t = 0; t < 3; t++) fe_to_bytes(&FR, combined + 96 * j + 32 * t, &comb[t]);
    fe rj; fe_from_bytes(&FR, &rj, challenges + 32 * j);
    for (size_t t = 0; t < ntab; t++) bind_fe(T[t], h, &rj, threads);                       /* sumcheck.rs:289-299 */
    cur = h;
  }
  if (finals) for (size_t t = 0; t < ntab; t++) fe_to_bytes(&FR, finals + 32 * t, &T[t][0]);
  for (size_t t = 0; t < ntab; t++) free(T[t]);
  free(T); free(co);
}
/* unipoly.rs:28-59 — same expression order as the reference */
void orc_unipoly_from_evals(const uint8_t* evals, size_t n, uint8_t* coeffs) {
  fe e[4], two, six, two_inv, six_inv; for (size_t i = 0; i < n; i++) fe_from_bytes(&FR, &e[i], evals + 32 * i);
  fe_from_u64(&FR, &two, 2); fe_from_u64(&FR, &six, 6); fe_inv(&FR, &two_inv, &two); fe_inv(&FR, &six_inv, &six);
  if (n == 3) {
    fe c = e[0], a, b; fe_sub(&FR, &a, &e[2], &e[1]); fe_sub(&FR, &a, &a, &e[1]); fe_add(&FR, &a, &a, &c); fe_mul(&FR, &a, &a, &two_inv);
    fe_sub(&FR, &b, &e[1], &c); fe_sub(&FR, &b, &b, &a);
    fe_to_bytes(&FR, coeffs, &c); fe_to_bytes(&FR, coeffs + 32, &b); fe_to_bytes(&FR, coeffs + 64, &a);
  } else {
    fe d = e[0], a, b, c, t;
    fe_sub(&FR, &a, &e[3], &e[2]); fe_sub(&FR, &a, &a, &e[2]); fe_sub(&FR, &a, &a, &e[2]); fe_add(&FR, &a, &a, &e[1]); fe_add(&FR, &a, &a, &e[1]); fe_add(&FR, &a, &a, &e[1]); fe_sub(&FR, &a, &a, &e[0]); fe_mul(&FR, &a, &a, &six_inv);
    fe_add(&FR, &b, &e[0], &e[0]); for (int k = 0; k < 5; k++) fe_sub(&FR, &b, &b, &e[1]); for (int k = 0; k < 4; k++) fe_add(&FR, &b, &b, &e[2]); fe_sub(&FR, &b, &b, &e[3]); fe_mul(&FR, &b, &b, &two_inv);
    fe_sub(&FR, &c, &e[1], &d); fe_sub(&FR, &c, &c, &a); fe_sub(&FR, &c, &c, &b); (void)t;
    fe_to_bytes(&FR, coeffs, &d); fe_to_bytes(&FR, coeffs + 32, &c); fe_to_bytes(&FR, coeffs + 64, &b); fe_to_bytes(&FR, coeffs + 96, &a);
  }
}
/* unipoly.rs:74-82 */
void orc_unipoly_eval(const uint8_t* coeffs, size_t n, const uint8_t r[32], uint8_t out[32]) {
  fe ev, pw, rr, c, t; fe_from_bytes(&FR, &ev, coeffs); fe_from_bytes(&FR, &rr, r); pw = rr;
  for (size_t i = 1; i < n; i++) { fe_from_bytes(&FR, &c, coeffs + 32 * i); fe_mul(&FR, &t, &pw, &c); fe_add(&FR, &ev, &ev, &t); fe_mul(&FR, &pw, &pw, &rr); }
  fe_to_bytes(&FR, out, &ev);
}

/* ------------------------------------------------------------------ BulletReductionProof::prove (nizk/bullet.rs:41-126)
 * The reference draws u_i from the transcript after absorbing L_i, R_i (bullet.rs:80-83); the transcript is outside the
 * path, so the challenges are an input here.  Everything else follows the reference line by line, with plain
 * double-and-add scalar multiplications (the mathematical definition). */
static void jac_mul_scalar_bytes(jac* o, const aff* p, const uint8_t k[32]) { jac j; fe s; jac_from_aff(&j, p); load_le(&s, k); jac_mul_int(o, &j, s.v); }
static void msm_def(jac* acc, const fe* k_mont, const aff* P, size_t n) {   /* sum k_i P_i, k in Montgomery form */
  jac_set_inf(acc);
  for (size_t i = 0; i < n; i++) { fe kc; fe_from_mont(&FR, &kc, &k_mont[i]); jac p, t; jac_from_aff(&p, &P[i]); jac_mul_int(&t, &p, kc.v); jac_add(acc, acc, &t); }
}
static void fr_dot_mont(fe* o, const fe* a, const fe* b, size_t n) { fe acc = {{0, 0, 0, 0}}, t; for (size_t i = 0; i < n; i++) { fe_mul(&FR, &t, &a[i], &b[i]); fe_add(&FR, &acc, &acc, &t); } *o = acc; }
void orc_bullet_prove(const uint8_t* G_xy, const uint8_t Q_xy[64], const uint8_t H_xy[64], const uint8_t* a_in, const uint8_t* b_in, size_t n,
                      const uint8_t blind[32], const uint8_t* blinds_vec /* lg n x (L, R) x 32 */, const uint8_t* us /* lg n x 32 */,
                      uint8_t* L_vec, uint8_t* R_vec, uint8_t Gamma[64], uint8_t a_hat[32], uint8_t b_hat[32], uint8_t g_hat[64], uint8_t blind_hat[32]) {
  aff* G = (aff*)malloc(sizeof(aff) * n); fe* a = (fe*)malloc(sizeof(fe) * n); fe* b = (fe*)malloc(sizeof(fe) * n);
  for (size_t i = 0; i < n; i++) { aff_from_bytes(&G[i], G_xy + 64 * i); fe_from_bytes(&FR, &a[i], a_in + 32 * i); fe_from_bytes(&FR, &b[i], b_in + 32 * i); }
  aff Qp, Hp; aff_from_bytes(&Qp, Q_xy); aff_from_bytes(&Hp, H_xy);
  fe bG; fe_from_bytes(&FR, &bG, blind);
  /* Gamma = MSM(a, G) + <a,b> Q + blind H  (bullet.rs:58-60) */
  { jac g, t; fe d; uint8_t db[32]; msm_def(&g, a, G, n); fr_dot_mont(&d, a, b, n); fe_to_bytes(&FR, db, &d);
    jac_mul_scalar_bytes(&t, &Qp, db); jac_add(&g, &g, &t); jac_mul_scalar_bytes(&t, &Hp, blind); jac_add(&g, &g, &t); jac_to_bytes(Gamma, &g); }
  size_t round = 0;
  while (n > 1) {
    n /= 2;
    fe cL, cR; fr_dot_mont(&cL, a, b + n, n); fr_dot_mont(&cR, a + n, b, n);          /* bullet.rs:72-73 */
    const uint8_t* bl = blinds_vec + 64 * round; const uint8_t* br = bl + 32;
    jac L, R, t; uint8_t sb[32];
    msm_def(&L, a, G + n, n); fe_to_bytes(&FR, sb, &cL); jac_mul_scalar_bytes(&t, &Qp, sb); jac_add(&L, &L, &t); jac_mul_scalar_bytes(&t, &Hp, bl); jac_add(&L, &L, &t);
    msm_def(&R, a + n, G, n); fe_to_bytes(&FR, sb, &cR); jac_mul_scalar_bytes(&t, &Qp, sb); jac_add(&R, &R, &t); jac_mul_scalar_bytes(&t, &Hp, br); jac_add(&R, &R, &t);
    jac_to_bytes(L_vec + 64 * round, &L); jac_to_bytes(R_vec + 64 * round, &R);
    fe u, ui; fe_from_bytes(&FR, &u, us + 32 * round); fe_inv(&FR, &ui, &u);
    uint8_t ub[32], uib[32]; fe_to_bytes(&FR, ub, &u); fe_to_bytes(&FR, uib, &ui);
    for (size_t i = 0; i < n; i++) {                                                   /* bullet.rs:87-106 */
      jac x, y; jac_mul_scalar_bytes(&x, &G[i], uib); jac_mul_scalar_bytes(&y, &G[i + n], ub); jac_add(&x, &x, &y); jac_to_aff(&G[i], &x);
      fe p, q; fe_mul(&FR, &p, &u, &a[i]); fe_mul(&FR, &q, &ui, &a[i + n]); fe_add(&FR, &a[i], &p, &q);
      fe_mul(&FR, &p, &ui, &b[i]); fe_mul(&FR, &q, &u, &b[i + n]); fe_add(&FR, &b[i], &p, &q);
    }
    fe blm, brm, uu, t1; fe_from_bytes(&FR, &blm, bl); fe_from_bytes(&FR, &brm, br);   /* bullet.rs:108 */
    fe_mul(&FR, &uu, &u, &u); fe_mul(&FR, &t1, &uu, &blm); fe_add(&FR, &bG, &bG, &t1);
    fe_mul(&FR, &uu, &ui, &ui); fe_mul(&FR, &t1, &uu, &brm); fe_add(&FR, &bG, &bG, &t1);
    round++;
  }
  fe_to_bytes(&FR, a_hat, &a[0]); fe_to_bytes(&FR, b_hat, &b[0]); aff_to_bytes(g_hat, &G[0]); fe_to_bytes(&FR, blind_hat, &bG);
  free(G); free(a); free(b);
}
