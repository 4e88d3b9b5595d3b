// g1.cuh — BN254 G1 (y^2 = x^3 + 3 over Fq) point arithmetic for gfx950.
//
// Stands in for ark-ec's short-Weierstrass group law behind `GroupElement(G1Projective)` and
// `G1Affine` (reference src/group.rs:11-13,143-158,171-175).  Bucket accumulators use extended
// Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): the mixed add is 8M+2S versus
// 7M+4S (11 products) for Jacobian, and the VALU cost is products (fp.cuh), not registers.
//
// Every formula handles the degenerate inputs explicitly — identity, P+P, P+(-P) — because the
// reference's generator derivation makes ~66 % of its bases literally equal to G
// (group.rs:110-131 falls back to 1*G when the hash is >= r), so equal and opposite
// operands are the COMMON case on the real workload, not a corner.
#pragma once
#include "fp.cuh"

namespace sbn {

struct alignas(16) Affine {   // 64 B; infinity is encoded as x = y = 0 ((0,0) is not on the curve)
  Fq x, y;
};
struct alignas(16) XYZZ {     // 128 B; infinity <=> ZZ == 0
  Fq X, Y, ZZ, ZZZ;
};

__device__ __forceinline__ bool aff_is_inf(const Affine& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
__device__ __forceinline__ bool xyzz_is_inf(const XYZZ& p) { return fe_is_zero(p.ZZ); }
__device__ __forceinline__ XYZZ xyzz_inf() { XYZZ r; r.X = fe_zero<FqP>(); r.Y = fe_zero<FqP>(); r.ZZ = fe_zero<FqP>(); r.ZZZ = fe_zero<FqP>(); return r; }
__device__ __forceinline__ XYZZ xyzz_from_affine(const Affine& p) {
  XYZZ r; r.X = p.x; r.Y = p.y;
  if (aff_is_inf(p)) { r.ZZ = fe_zero<FqP>(); r.ZZZ = fe_zero<FqP>(); } else { r.ZZ = fe_one<FqP>(); r.ZZZ = fe_one<FqP>(); }
  return r;
}

__device__ __forceinline__ Affine aff_load(const void* p) {
  Affine a; a.x = fe_load<FqP>(p); a.y = fe_load<FqP>(reinterpret_cast<const uint8_t*>(p) + 32); return a;
}
__device__ __forceinline__ void aff_store(void* p, const Affine& a) {
  fe_store<FqP>(p, a.x); fe_store<FqP>(reinterpret_cast<uint8_t*>(p) + 32, a.y);
}
__device__ __forceinline__ XYZZ xyzz_load(const void* p) {
  const uint8_t* q = reinterpret_cast<const uint8_t*>(p);
  XYZZ r; r.X = fe_load<FqP>(q); r.Y = fe_load<FqP>(q + 32); r.ZZ = fe_load<FqP>(q + 64); r.ZZZ = fe_load<FqP>(q + 96); return r;
}
__device__ __forceinline__ void xyzz_store(void* p, const XYZZ& a) {
  uint8_t* q = reinterpret_cast<uint8_t*>(p);
  fe_store<FqP>(q, a.X); fe_store<FqP>(q + 32, a.Y); fe_store<FqP>(q + 64, a.ZZ); fe_store<FqP>(q + 96, a.ZZZ);
}

// 2*P for an affine P != infinity (y != 0 always holds on this curve: no 2-torsion in G1).
// Inlined on purpose: an out-of-line callee takes its 64/128-byte operand by reference, which forces the caller to keep a copy
// in scratch memory on EVERY loop iteration even though the doubling branch is rare (measured: ~1 GB of scratch stores per
// 2^20 MSM in the accumulate kernel).
__device__ __forceinline__ XYZZ xyzz_dbl_affine(const Affine& p) {
  Fq U = fe_dbl(p.y), V = fe_sqr(U), W = fe_mul(U, V), S = fe_mul(p.x, V);
  Fq xx = fe_sqr(p.x), M3 = fe_add(fe_dbl(xx), xx);
  XYZZ r;
  r.X = fe_sub(fe_sub(fe_sqr(M3), S), S);
  r.Y = fe_sub(fe_mul(M3, fe_sub(S, r.X)), fe_mul(W, p.y));
  r.ZZ = V; r.ZZZ = W;
  return r;
}
// 2*P, XYZZ
__device__ __forceinline__ XYZZ xyzz_dbl(const XYZZ& p) {
  if (xyzz_is_inf(p)) return p;
  Fq U = fe_dbl(p.Y), V = fe_sqr(U), W = fe_mul(U, V), S = fe_mul(p.X, V);
  Fq xx = fe_sqr(p.X), M3 = fe_add(fe_dbl(xx), xx);
  XYZZ r;
  r.X = fe_sub(fe_sub(fe_sqr(M3), S), S);
  r.Y = fe_sub(fe_mul(M3, fe_sub(S, r.X)), fe_mul(W, p.Y));
  r.ZZ = fe_mul(V, p.ZZ); r.ZZZ = fe_mul(W, p.ZZZ);
  return r;
}

// acc += q (q affine, possibly infinity); neg => acc -= q.  8M + 2S on the common path.
__device__ __forceinline__ void xyzz_madd(XYZZ& acc, const Affine& q_in, bool neg) {
  if (aff_is_inf(q_in)) return;
  Affine q = q_in;
  if (neg) q.y = fe_neg(q.y);
  if (xyzz_is_inf(acc)) { acc.X = q.x; acc.Y = q.y; acc.ZZ = fe_one<FqP>(); acc.ZZZ = acc.ZZ; return; }
  Fq U2 = fe_mul(q.x, acc.ZZ), S2 = fe_mul(q.y, acc.ZZZ);
  Fq P = fe_sub(U2, acc.X), R = fe_sub(S2, acc.Y);
  if (fe_is_zero(P)) {                         // same x: doubling or cancellation
    if (fe_is_zero(R)) acc = xyzz_dbl_affine(q); else acc = xyzz_inf();
    return;
  }
  Fq PP = fe_sqr(P), PPP = fe_mul(P, PP), Q = fe_mul(acc.X, PP);
  Fq X3 = fe_sub(fe_sub(fe_sub(fe_sqr(R), PPP), Q), Q);
  Fq Y3 = fe_sub(fe_mul(R, fe_sub(Q, X3)), fe_mul(acc.Y, PPP));
  acc.X = X3; acc.Y = Y3; acc.ZZ = fe_mul(acc.ZZ, PP); acc.ZZZ = fe_mul(acc.ZZZ, PPP);
}

// a + b, both XYZZ.  12M + 2S on the common path.  The inlined form is for the latency-bound reduction kernels (a call
// passes both 128-byte operands through scratch); xyzz_add is the out-of-line copy for code-size-sensitive callers.
__device__ __forceinline__ XYZZ xyzz_add_inl(const XYZZ& a, const XYZZ& b) {
  if (xyzz_is_inf(a)) return b;
  if (xyzz_is_inf(b)) return a;
  Fq U1 = fe_mul(a.X, b.ZZ), U2 = fe_mul(b.X, a.ZZ), S1 = fe_mul(a.Y, b.ZZZ), S2 = fe_mul(b.Y, a.ZZZ);
  Fq P = fe_sub(U2, U1), R = fe_sub(S2, S1);
  if (fe_is_zero(P)) {
    if (fe_is_zero(R)) return xyzz_dbl(a);
    return xyzz_inf();
  }
  Fq PP = fe_sqr(P), PPP = fe_mul(P, PP), Q = fe_mul(U1, PP);
  XYZZ r;
  r.X = fe_sub(fe_sub(fe_sub(fe_sqr(R), PPP), Q), Q);
  r.Y = fe_sub(fe_mul(R, fe_sub(Q, r.X)), fe_mul(S1, PPP));
  r.ZZ = fe_mul(fe_mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = fe_mul(fe_mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}
__device__ __noinline__ XYZZ xyzz_add(const XYZZ& a, const XYZZ& b) { return xyzz_add_inl(a, b); }

// canonical affine (Montgomery coordinates) from XYZZ: one inversion
__device__ __noinline__ Affine xyzz_to_affine(const XYZZ& p) {
  Affine r;
  if (xyzz_is_inf(p)) { r.x = fe_zero<FqP>(); r.y = fe_zero<FqP>(); return r; }
  Fq I = fe_inv(fe_mul(p.ZZ, p.ZZZ));           // 1/(ZZ*ZZZ)
  r.x = fe_mul(p.X, fe_mul(I, p.ZZZ));          // X / ZZ
  r.y = fe_mul(p.Y, fe_mul(I, p.ZZ));           // Y / ZZZ
  return r;
}

// lane <- lane+delta exchange of a whole point inside one 64-wide wavefront
__device__ __forceinline__ XYZZ xyzz_shfl_down(const XYZZ& p, int delta) {
  XYZZ r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.X.v[i] = __shfl_down(p.X.v[i], delta, 64); r.Y.v[i] = __shfl_down(p.Y.v[i], delta, 64);
    r.ZZ.v[i] = __shfl_down(p.ZZ.v[i], delta, 64); r.ZZZ.v[i] = __shfl_down(p.ZZZ.v[i], delta, 64);
  }
  return r;
}

__device__ __forceinline__ XYZZ xyzz_shfl_xor(const XYZZ& p, int mask) {
  XYZZ r;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r.X.v[i] = __shfl_xor(p.X.v[i], mask, 64); r.Y.v[i] = __shfl_xor(p.Y.v[i], mask, 64);
    r.ZZ.v[i] = __shfl_xor(p.ZZ.v[i], mask, 64); r.ZZZ.v[i] = __shfl_xor(p.ZZZ.v[i], mask, 64);
  }
  return r;
}

}  // namespace sbn
