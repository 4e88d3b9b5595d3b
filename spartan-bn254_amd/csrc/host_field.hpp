// host_field.hpp — host-side (CPU) BN254 Fq / G1 helpers used by the product's launch code.
//
// What runs here is the part of Pippenger that is a strictly sequential chain of ~254 doublings
// (combining the per-window sums: sum_w 2^(c*w) * S_w).  A lone GPU lane needs ~1 us per modular
// product (fp.cuh), i.e. ~2 ms for that chain; one host core does it in ~0.1 ms.  This is product
// code (4 x 64-bit limbs, Jacobian) and shares nothing with oracle/ — the oracle stays a checker.
// Reference semantics: group.rs:171-175 (the sum is returned as a group element; we hand back its
// canonical affine coordinates).
#pragma once
#include <stdint.h>
#include <string.h>

namespace sbn_host {

typedef unsigned __int128 u128;
struct Fq { uint64_t v[4]; };

static const uint64_t QP[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t QONE[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
static const uint64_t QR2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full};
static const uint64_t QNINV = 0x87d20782e4866389ull;

static inline bool geq_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > QP[i]) return true; if (a[i] < QP[i]) return false; }
  return true;
}
static inline void sub_p(uint64_t a[4]) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - QP[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 127); }
}
static inline bool is_zero(const Fq& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static inline bool eq(const Fq& a, const Fq& b) { return memcmp(a.v, b.v, 32) == 0; }
static inline Fq add(const Fq& a, const Fq& b) {
  Fq r; uint64_t c = 0;
  for (int i = 0; i < 4; i++) { u128 s = (u128)a.v[i] + b.v[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  if (c || geq_p(r.v)) sub_p(r.v);
  return r;
}
static inline Fq sub(const Fq& a, const Fq& b) {
  Fq r; uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.v[i] - b.v[i] - br; r.v[i] = (uint64_t)d; br = (uint64_t)(d >> 127); }
  if (br) { uint64_t c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)r.v[i] + QP[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
  return r;
}
// interleaved (CIOS) Montgomery product
static inline Fq mul(const Fq& a, const Fq& b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 4; j++) { u128 s = (u128)a.v[j] * b.v[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    u128 s = (u128)t[4] + c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
    uint64_t m = t[0] * QNINV;
    s = (u128)m * QP[0] + t[0]; c = (uint64_t)(s >> 64);
    for (int j = 1; j < 4; j++) { s = (u128)m * QP[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
  }
  Fq r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_p(r.v)) sub_p(r.v);
  return r;
}
static inline Fq sqr(const Fq& a) { return mul(a, a); }
static inline Fq dbl(const Fq& a) { return add(a, a); }
static inline Fq one() { Fq r; memcpy(r.v, QONE, 32); return r; }
static inline Fq zero() { Fq r = {{0, 0, 0, 0}}; return r; }
static inline Fq inv_fermat(const Fq& a) {
  uint64_t e[4] = {QP[0] - 2, QP[1], QP[2], QP[3]};
  Fq acc = one();
  for (int i = 255; i >= 0; i--) { acc = sqr(acc); if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, a); }
  return acc;
}
// 1 / x of an integer 0 < x < m (m odd, below 2^255), as an integer: the binary extended Euclid — u, v shrink by shifts and
// subtractions while b x = u and c x = v (mod m) are kept.  ~2 x 254 steps on four limbs: 2-3 us against ~15-20 us for Fermat's 380
// Montgomery products.  Variable time: its operands are public (coordinates of commitments, transcript challenges).
static inline void inv_mod_odd(const uint64_t x[4], const uint64_t m[4], uint64_t out[4]) {
  auto is_one = [](const uint64_t a[4]) { return a[0] == 1 && (a[1] | a[2] | a[3]) == 0; };
  auto shr1 = [](uint64_t a[4], uint64_t top) { a[0] = (a[0] >> 1) | (a[1] << 63); a[1] = (a[1] >> 1) | (a[2] << 63); a[2] = (a[2] >> 1) | (a[3] << 63); a[3] = (a[3] >> 1) | (top << 63); };
  auto halve = [&](uint64_t a[4]) {                       // a / 2 mod m for a < m: (a + m) / 2 when a is odd
    uint64_t c = 0;
    if (a[0] & 1) for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + m[i] + c; a[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    shr1(a, c);
  };
  auto lt = [](const uint64_t a[4], const uint64_t b[4]) { for (int i = 3; i >= 0; i--) { if (a[i] < b[i]) return true; if (a[i] > b[i]) return false; } return false; };
  auto sub_raw = [](uint64_t a[4], const uint64_t b[4]) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 127); } };
  auto sub_mod = [&](uint64_t a[4], const uint64_t b[4]) {        // a <- a - b mod m, both below m
    if (lt(a, b)) { uint64_t c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + m[i] + c; a[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
    sub_raw(a, b);
  };
  uint64_t u[4] = {x[0], x[1], x[2], x[3]}, v[4] = {m[0], m[1], m[2], m[3]}, b[4] = {1, 0, 0, 0}, c[4] = {0, 0, 0, 0};
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1)) { shr1(u, 0); halve(b); }
    while (!(v[0] & 1)) { shr1(v, 0); halve(c); }
    if (lt(v, u)) { sub_raw(u, v); sub_mod(b, c); } else { sub_raw(v, u); sub_mod(c, b); }
  }
  memcpy(out, is_one(u) ? b : c, 32);
}
// Montgomery form in, Montgomery form out: the integer inverse of a R is a^-1 R^-1; two products with R^2 bring it to a^-1 R.  0 -> 0.
static inline Fq inv(const Fq& a) {
  if (is_zero(a)) return a;
  Fq x, r2; memcpy(r2.v, QR2, 32);
  inv_mod_odd(a.v, QP, x.v);
  return mul(mul(x, r2), r2);
}
static inline Fq from_mont(const Fq& a) { Fq o = {{1, 0, 0, 0}}; return mul(a, o); }
static inline Fq to_mont(const Fq& a) { Fq r2; memcpy(r2.v, QR2, 32); return mul(a, r2); }

// XYZZ point, same layout as the device's (4 x 32 B, Montgomery limbs little-endian)
struct Pt { Fq X, Y, ZZ, ZZZ; };
// The device's field layer (fp.cuh) keeps Montgomery form with R = 2^261 and stores lazy representatives (any non-negative
// integer below 2^256 congruent to x * 2^261); this file's CIOS uses R = 2^256 and canonical values.  One product with 2^251
// converts: (x 2^261 + k p) * 2^251 * 2^-256 = x * 2^256 (mod p), and the CIOS result is canonical for any 256-bit left operand.
static const uint64_t QK251[4] = {0, 0, 0, 0x0800000000000000ull};
static inline Fq from_device(const Fq& a) { Fq k; memcpy(k.v, QK251, 32); return mul(a, k); }
static inline bool is_inf(const Pt& p) { return is_zero(p.ZZ); }
static inline Pt pt_from_device(const Pt& d) { Pt p; p.X = from_device(d.X); p.Y = from_device(d.Y); p.ZZ = from_device(d.ZZ); p.ZZZ = from_device(d.ZZZ); return p; }
static inline Pt inf() { Pt p; p.X = zero(); p.Y = zero(); p.ZZ = zero(); p.ZZZ = zero(); return p; }
static inline Pt pdbl(const Pt& p) {
  if (is_inf(p)) return p;
  Fq U = dbl(p.Y), V = sqr(U), W = mul(U, V), S = mul(p.X, V), xx = sqr(p.X), M3 = add(dbl(xx), xx);
  Pt r; r.X = sub(sub(sqr(M3), S), S); r.Y = sub(mul(M3, sub(S, r.X)), mul(W, p.Y)); r.ZZ = mul(V, p.ZZ); r.ZZZ = mul(W, p.ZZZ);
  return r;
}
static inline Pt padd(const Pt& a, const Pt& b) {
  if (is_inf(a)) return b;
  if (is_inf(b)) return a;
  Fq U1 = mul(a.X, b.ZZ), U2 = mul(b.X, a.ZZ), S1 = mul(a.Y, b.ZZZ), S2 = mul(b.Y, a.ZZZ);
  Fq P = sub(U2, U1), R = sub(S2, S1);
  if (is_zero(P)) return is_zero(R) ? pdbl(a) : inf();
  Fq PP = sqr(P), PPP = mul(P, PP), Q = mul(U1, PP);
  Pt r; r.X = sub(sub(sub(sqr(R), PPP), Q), Q); r.Y = sub(mul(R, sub(Q, r.X)), mul(S1, PPP));
  r.ZZ = mul(mul(a.ZZ, b.ZZ), PP); r.ZZZ = mul(mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}
// canonical little-endian x||y (64 B), all-zero for infinity
static inline void to_affine_bytes(const Pt& p, uint8_t out[64], int* is_inf_out) {
  if (is_inf(p)) { memset(out, 0, 64); if (is_inf_out) *is_inf_out = 1; return; }
  Fq I = inv(mul(p.ZZ, p.ZZZ));
  Fq x = from_mont(mul(p.X, mul(I, p.ZZZ))), y = from_mont(mul(p.Y, mul(I, p.ZZ)));
  memcpy(out, x.v, 32); memcpy(out + 32, y.v, 32);   // little-endian host
  if (is_inf_out) *is_inf_out = 0;
}
// two points with ONE inversion (Montgomery's trick): a bullet round returns L and R together
static inline void to_affine_bytes2(const Pt& p, const Pt& q, uint8_t outp[64], int* p_inf, uint8_t outq[64], int* q_inf) {
  if (is_inf(p) || is_inf(q)) { to_affine_bytes(p, outp, p_inf); to_affine_bytes(q, outq, q_inf); return; }
  const Fq zp = mul(p.ZZ, p.ZZZ), zq = mul(q.ZZ, q.ZZZ);
  const Fq I = inv(mul(zp, zq));
  const Fq Ip = mul(I, zq), Iq = mul(I, zp);
  const Fq xp = from_mont(mul(p.X, mul(Ip, p.ZZZ))), yp = from_mont(mul(p.Y, mul(Ip, p.ZZ)));
  const Fq xq = from_mont(mul(q.X, mul(Iq, q.ZZZ))), yq = from_mont(mul(q.Y, mul(Iq, q.ZZ)));
  memcpy(outp, xp.v, 32); memcpy(outp + 32, yp.v, 32); memcpy(outq, xq.v, 32); memcpy(outq + 32, yq.v, 32);
  if (p_inf) *p_inf = 0;
  if (q_inf) *q_inf = 0;
}
// sum_w 2^(c*w) * S[w], w = 0..W-1 (Horner from the top window)
static inline Pt combine_windows(const Pt* S, int W, int c) {
  Pt acc = inf();
  for (int w = W - 1; w >= 0; w--) {
    for (int d = 0; d < c; d++) acc = pdbl(acc);
    acc = padd(acc, S[w]);
  }
  return acc;
}

// ---- Fr (scalar field) on the host: only what UniPoly::from_evals / evaluate need (unipoly.rs:28-82) -----------------
namespace fr {
static const uint64_t P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static inline bool geq(const uint64_t a[4]) { for (int i = 3; i >= 0; i--) { if (a[i] > P[i]) return true; if (a[i] < P[i]) return false; } return true; }
struct El { uint64_t v[4]; };                      // canonical integers (no Montgomery form: a handful of operations per round)
static inline El add(const El& a, const El& b) {
  El r; uint64_t c = 0;
  for (int i = 0; i < 4; i++) { u128 s = (u128)a.v[i] + b.v[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  if (c || geq(r.v)) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r.v[i] - P[i] - br; r.v[i] = (uint64_t)d; br = (uint64_t)(d >> 127); } }
  return r;
}
static inline El sub(const El& a, const El& b) {
  El r; uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a.v[i] - b.v[i] - br; r.v[i] = (uint64_t)d; br = (uint64_t)(d >> 127); }
  if (br) { uint64_t c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)r.v[i] + P[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
  return r;
}
// a*b mod r by shift-and-add over the bits of b (256 modular doublings: fine for 3-4 values per sumcheck round)
static inline El mul(const El& a, const El& b) {
  El acc = {{0, 0, 0, 0}};
  for (int i = 255; i >= 0; i--) { acc = add(acc, acc); if ((b.v[i >> 6] >> (i & 63)) & 1) acc = add(acc, a); }
  return acc;
}
static inline El from_u64(uint64_t x) { El r = {{x, 0, 0, 0}}; return r; }
static inline El inv_small(uint64_t k) {           // 1/k for k in {2, 6}: (r*m + 1)/k with the m that makes it divisible
  // r = 1 mod 2 and r = 1 mod 6 (r - 1 is divisible by 2^28 * 3), so (r*(k-1) + 1) / k is an integer < r and k times it is 1 mod r
  u128 carry = 1; uint64_t t[5];
  for (int i = 0; i < 4; i++) { u128 s = (u128)P[i] * (k - 1) + carry; t[i] = (uint64_t)s; carry = s >> 64; }
  t[4] = (uint64_t)carry;
  El q; u128 rem = 0;
  for (int i = 4; i >= 0; i--) { u128 cur = (rem << 64) | t[i]; uint64_t d = (uint64_t)(cur / k); rem = cur % k; if (i < 4) q.v[i] = d; }
  return q;
}
// Montgomery product a * b * 2^-256 mod r (CIOS, 4 x 64-bit limbs): what the stateful sumcheck's host side uses to combine the
// "seq" instances' sums with coeffs and to recover the final claims (a few dozen products per round; the shift-and-add mul above
// costs microseconds each).  mmul(to_m(a), b) = a * b for canonical a, b.
static const uint64_t NINV = 0xc2e1f593efffffffull;
static const uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};   // 2^512 mod r
static inline El mmul(const El& a, const El& b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 4; j++) { u128 s = (u128)a.v[j] * b.v[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    u128 s = (u128)t[4] + c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
    const uint64_t m = t[0] * NINV;
    s = (u128)m * P[0] + t[0]; c = (uint64_t)(s >> 64);
    for (int j = 1; j < 4; j++) { s = (u128)m * P[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
  }
  El r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq(r.v)) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r.v[i] - P[i] - br; r.v[i] = (uint64_t)d; br = (uint64_t)(d >> 127); } }
  return r;
}
static inline El to_m(const El& a) { El r2; memcpy(r2.v, R2, 32); return mmul(a, r2); }                 // a * 2^256
static inline El from_m(const El& a) { return mmul(a, from_u64(1)); }
static inline bool is_zero(const El& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
// 1 / a, canonical in and out: Fermat's exponentiation (kept as the cross-check of the tests) and the binary Euclid above
// (inv_mod_odd; the operands here are transcript challenges and batching coefficients, public values).  0 has no inverse: returns 0.
static inline El inv_fermat(const El& a) {
  const uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
  const El am = to_m(a);
  El acc = to_m(from_u64(1));
  for (int i = 255; i >= 0; i--) { acc = mmul(acc, acc); if ((e[i >> 6] >> (i & 63)) & 1) acc = mmul(acc, am); }
  return from_m(acc);
}
static inline El inv(const El& a) {
  if (is_zero(a)) return a;
  El r; inv_mod_odd(a.v, P, r.v);
  return r;
}
// canonical x -> the device's table representation x * 2^261 mod r (fp.cuh's Montgomery domain), canonical
static inline El to_dev_mont(const El& x) {
  static const El K = [] { El k; memcpy(k.v, R2, 32); for (int i = 0; i < 5; i++) k = add(k, k); return k; }();    // 2^517 mod r
  return mmul(x, K);
}
}  // namespace fr

}  // namespace sbn_host
