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
//
// Field values are lazy (fp.cuh): coordinates are kept NORMALISED and NON-NEGATIVE with these value ranges, which the two hot
// formulas (xyzz_madd, xyzz_add_inl: fp.cuh's unsigned fast path) rely on and re-establish:
//      X in [0, 5.2p)    Y in [0, 3.2p)    ZZ, ZZZ in [0, 1.2p)    affine x, y in [0, p) (tables are canonical)
// Infinity is the exact all-zero limb pattern of ZZ (XYZZ) or of x and y (affine); no computed non-identity point ever has it
// (its ZZ is a product of non-zero elements), and the memory formats below keep zero as zero.
#pragma once
#include "fp.cuh"

namespace sbn {

struct Affine {   // memory: 64 B canonical x || y; infinity is encoded as x = y = 0 ((0,0) is not on the curve)
  Fq x, y;
};
struct XYZZ {     // memory: 128 B; infinity <=> ZZ == 0
  Fq X, Y, ZZ, ZZZ;
};

__device__ __forceinline__ bool aff_is_inf(const Affine& p) { return fe_is_zero_limbs(p.x) && fe_is_zero_limbs(p.y); }
__device__ __forceinline__ bool xyzz_is_inf(const XYZZ& p) { return fe_is_zero_limbs(p.ZZ); }
__device__ __forceinline__ XYZZ xyzz_inf() { XYZZ r; r.X = fe_zero<FqP>(); r.Y = fe_zero<FqP>(); r.ZZ = fe_zero<FqP>(); r.ZZZ = fe_zero<FqP>(); return r; }
__device__ __forceinline__ XYZZ xyzz_from_affine(const Affine& p) {
  XYZZ r; r.X = p.x; r.Y = p.y;
  if (aff_is_inf(p)) { r.ZZ = fe_zero<FqP>(); r.ZZZ = fe_zero<FqP>(); } else { r.ZZ = fe_one<FqP>(); r.ZZZ = fe_one<FqP>(); }
  return r;
}

__device__ __forceinline__ Affine aff_load(const void* p) {
  Affine a; a.x = fe_load<FqP>(p); a.y = fe_load<FqP>(reinterpret_cast<const uint8_t*>(p) + 32); return a;
}
__device__ __forceinline__ void aff_store(void* p, const Affine& a) {      // canonical (tables, results)
  fe_store<FqP>(p, a.x); fe_store<FqP>(reinterpret_cast<uint8_t*>(p) + 32, a.y);
}
__device__ __forceinline__ XYZZ xyzz_load(const void* p) {
  const uint8_t* q = reinterpret_cast<const uint8_t*>(p);
  XYZZ r; r.X = fe_load<FqP>(q); r.Y = fe_load<FqP>(q + 32); r.ZZ = fe_load<FqP>(q + 64); r.ZZZ = fe_load<FqP>(q + 96); return r;
}
// Intermediate sums (buckets, partial sums) are stored as they are — non-negative representatives below 2^256 = 5.29 p, not
// canonical ones (the sign masks below are a safety net for values that came through the signed, rare-path formulas).
__device__ __forceinline__ void xyzz_store(void* p, const XYZZ& a) {
  uint8_t* q = reinterpret_cast<uint8_t*>(p);
  fe_store_packed<FqP>(q, fe_fix_nonneg<FqP, 4>(fe_norm(a.X))); fe_store_packed<FqP>(q + 32, fe_fix_nonneg<FqP, 2>(fe_norm(a.Y)));
  fe_store_packed<FqP>(q + 64, fe_fix_nonneg<FqP, 1>(fe_norm(a.ZZ))); fe_store_packed<FqP>(q + 96, fe_fix_nonneg<FqP, 1>(fe_norm(a.ZZZ)));
}

// 2*P for an affine P != infinity (y != 0 always holds on this curve: no 2-torsion in G1).  Rare path: the safe (signed,
// normalising) operations, then the coordinates are brought back to the non-negative form the hot formulas expect.
// Inlined on purpose: an out-of-line callee takes its operand by reference, which forces the caller to keep a copy in scratch memory.
__device__ __forceinline__ XYZZ xyzz_dbl_affine(const Affine& p) {
  const Fq U = fe_dbl(p.y), V = fe_sqr(U), W = fe_mul(U, V), S = fe_mul(p.x, V);
  const Fq xx = fe_sqr(p.x), M3 = fe_add(fe_dbl(xx), xx);
  XYZZ r;
  const Fq X = fe_sub(fe_sub(fe_sqr(M3), S), S);                                  // (-2.2p, 1.1p)
  r.X = fe_fix_nonneg<FqP, 4>(X);
  r.Y = fe_fix_nonneg<FqP, 2>(fe_sub(fe_mul(M3, fe_sub(S, X)), fe_mul(W, p.y)));  // (-1.2p, 1.2p)
  r.ZZ = fe_fix_nonneg<FqP, 1>(V); r.ZZZ = fe_fix_nonneg<FqP, 1>(W);
  return r;
}
// 2*P, XYZZ
__device__ __forceinline__ XYZZ xyzz_dbl(const XYZZ& p) {
  if (xyzz_is_inf(p)) return p;
  const Fq U = fe_dbl(p.Y), V = fe_sqr(U), W = fe_mul(U, V), S = fe_mul(p.X, V);
  const Fq xx = fe_sqr(p.X), M3 = fe_add(fe_dbl(xx), xx);
  XYZZ r;
  const Fq X = fe_sub(fe_sub(fe_sqr(M3), S), S);
  r.X = fe_fix_nonneg<FqP, 4>(X);
  r.Y = fe_fix_nonneg<FqP, 2>(fe_sub(fe_mul(M3, fe_sub(S, X)), fe_mul(W, p.Y)));
  r.ZZ = fe_fix_nonneg<FqP, 1>(fe_mul(V, p.ZZ)); r.ZZZ = fe_fix_nonneg<FqP, 1>(fe_mul(W, p.ZZZ));
  return r;
}

// acc += q (q affine from a table: canonical, possibly infinity); neg => acc -= q.  8M + 2S on the common path, on the UNSIGNED
// fast path of fp.cuh: every difference carries an inflated multiple of p, so no limb is ever negative; four carry passes.
// Value ranges (re-established here): X in [0, 5.2p), Y in [0, 3.2p), ZZ, ZZZ in [0, 1.2p); all limbs normalised.
// The two infinity tests are exact limb tests (18 and 9 ORs) behind ONE-limb filters: limb 0 of a non-zero coordinate is zero once in
// 2^29, so the full tests run on the rare path only (tools/micro/maddvar.hip: the gather-fed addition 1.18 -> 1.30 x 10^10 /s at 3 waves
// per SIMD with the tests out of the way).
__device__ __forceinline__ void xyzz_madd(XYZZ& acc, const Affine& q_in, bool neg) {
  if ((q_in.x.v[0] | q_in.y.v[0]) == 0) { if (aff_is_inf(q_in)) return; }
  const Fq qy = neg ? fe_negb<FqP, 2>(q_in.y) : q_in.y;                    // 2p - y: limbs below 2^30
  if (acc.ZZ.v[0] == 0) { if (xyzz_is_inf(acc)) { acc.X = q_in.x; acc.Y = fe_normu(qy); acc.ZZ = fe_one<FqP>(); acc.ZZZ = acc.ZZ; return; } }
  const Fq U2 = fe_mulu(q_in.x, acc.ZZ), S2 = fe_mulu(qy, acc.ZZZ);        // [0, 1.2p)
  const Fq P = fe_normu(fe_subb<FqP, 6, 1>(U2, acc.X));                    // U2 - X + 6p in (0.8p, 7.2p)
  const Fq R = fe_normu(fe_subb<FqP, 4, 1>(S2, acc.Y));                    // S2 - Y + 4p in (0.8p, 5.2p)
  if (fe_maybe_zero(P) && fe_is_zero(P)) {                                 // same x: doubling or cancellation
    if (fe_is_zero(R)) { Affine q; q.x = q_in.x; q.y = fe_normu(qy); acc = xyzz_dbl_affine(q); } else acc = xyzz_inf();
    return;
  }
  const Fq PP = fe_squ(P), PPP = fe_mulu(P, PP), Q = fe_mulu(acc.X, PP);
  const Fq X3 = fe_normu(fe_subb<FqP, 4, 3>(fe_squ(R), fe_add_lazy(fe_add_lazy(PPP, Q), Q)));        // R^2 - PPP - 2Q + 4p in (0.4p, 5.2p)
  // Y3 = R (Q - X3) - Y PPP as ONE sum of two products with ONE Montgomery reduction (fp.cuh: Cols): R (Q - X3 + 6p) + (4p - Y) PPP.
  // Limb products: 2^29 x 2^30.6 and 2^30 x 2^29, nine per column each: 1.3e19 + the reduction's 2.6e18 < 2^64.  Value: at most
  // (5.2 * 7.2 + 4 * 1.2) p^2 / 2^261 + p < 1.3 p, normalised by the reduction itself (no carry pass, no second reduction).
  Cols cy; cols_zero(cy);
  cols_mac_lazy<FqP>(cy, R, fe_subb<FqP, 6, 1>(Q, X3)); cols_mac_lazy<FqP>(cy, PPP, fe_negb<FqP, 4>(acc.Y));
  acc.X = X3; acc.Y = cols_reduce<FqP>(cy); acc.ZZ = fe_mulu(acc.ZZ, PP); acc.ZZZ = fe_mulu(acc.ZZZ, PPP);
}

// a + b, both XYZZ (normalised, non-negative, ranges as above).  12M + 2S on the common path.  The inlined form is for the
// latency-bound reduction kernels (a call passes both operands through scratch); xyzz_add is the out-of-line copy.
__device__ __forceinline__ XYZZ xyzz_add_inl(const XYZZ& a, const XYZZ& b) {
  if (xyzz_is_inf(a)) return b;
  if (xyzz_is_inf(b)) return a;
  const Fq U1 = fe_mulu(a.X, b.ZZ), U2 = fe_mulu(b.X, a.ZZ), S1 = fe_mulu(a.Y, b.ZZZ), S2 = fe_mulu(b.Y, a.ZZZ);   // [0, 1.2p)
  const Fq P = fe_normu(fe_subb<FqP, 2, 1>(U2, U1)), R = fe_normu(fe_subb<FqP, 2, 1>(S2, S1));                  // (0.8p, 3.2p)
  if (fe_maybe_zero(P) && fe_is_zero(P)) {
    if (fe_is_zero(R)) return xyzz_dbl(a);
    return xyzz_inf();
  }
  const Fq PP = fe_squ(P), PPP = fe_mulu(P, PP), Q = fe_mulu(U1, PP);
  XYZZ r;
  r.X = fe_normu(fe_subb<FqP, 4, 3>(fe_squ(R), fe_add_lazy(fe_add_lazy(PPP, Q), Q)));
  Cols cy; cols_zero(cy);                                                            // R (Q - X3 + 6p) + (2p - S1) PPP, one reduction (see xyzz_madd); S1 < 1.2p
  cols_mac_lazy<FqP>(cy, R, fe_subb<FqP, 6, 1>(Q, r.X)); cols_mac_lazy<FqP>(cy, PPP, fe_negb<FqP, 2>(S1));
  r.Y = cols_reduce<FqP>(cy);
  r.ZZ = fe_mulu(fe_mulu(a.ZZ, b.ZZ), PP);
  r.ZZZ = fe_mulu(fe_mulu(a.ZZZ, b.ZZZ), PPP);
  return r;
}
__device__ __noinline__ XYZZ xyzz_add(const XYZZ& a, const XYZZ& b) { return xyzz_add_inl(a, b); }

// canonical affine (Montgomery coordinates in [0, p)) from XYZZ: one inversion
__device__ __noinline__ Affine xyzz_to_affine(const XYZZ& p) {
  Affine r;
  if (xyzz_is_inf(p)) { r.x = fe_zero<FqP>(); r.y = fe_zero<FqP>(); return r; }
  const Fq I = fe_inv(fe_mul(p.ZZ, p.ZZZ));        // 1/(ZZ*ZZZ)
  r.x = fe_canon(fe_mul(p.X, fe_mul(I, p.ZZZ)));   // X / ZZ
  r.y = fe_canon(fe_mul(p.Y, fe_mul(I, p.ZZ)));    // Y / ZZZ
  return r;
}

// lane <- lane+delta exchange of a whole point inside one 64-wide wavefront
__device__ __forceinline__ XYZZ xyzz_shfl_down(const XYZZ& p, int delta) {
  XYZZ r; r.X = fe_shfl_down(p.X, delta); r.Y = fe_shfl_down(p.Y, delta); r.ZZ = fe_shfl_down(p.ZZ, delta); r.ZZZ = fe_shfl_down(p.ZZZ, delta);
  return r;
}
__device__ __forceinline__ XYZZ xyzz_shfl_xor(const XYZZ& p, int mask) {
  XYZZ r; r.X = fe_shfl_xor(p.X, mask); r.Y = fe_shfl_xor(p.Y, mask); r.ZZ = fe_shfl_xor(p.ZZ, mask); r.ZZZ = fe_shfl_xor(p.ZZZ, mask);
  return r;
}

// ---- latency-shaped addition: the four lanes of a quad compute ONE a + b -----------------------------------------------------
// A tree sum on a few waves is a chain of dependent field products (0.54 us each on a lone wave: 7.6 us per full addition),
// while 3 of every 4 lanes sit idle from the second tree level on.  Here every lane of a quad holds both operands, each of the
// four product steps computes up to four DIFFERENT products of the formula (one per lane, operands picked by the lane's role),
// and the results are broadcast inside the quad (DPP): 4 dependent products instead of 14.  All four lanes return the sum.
// Same formula, same operand ranges and the same degenerate cases as xyzz_add_inl.  `role` = lane & 3.
// (force-inlined: an out-of-line copy would take its operands through scratch memory; block_sum_quad below has ONE call site)
__device__ __forceinline__ XYZZ xyzz_add_quad(const XYZZ& a, const XYZZ& b, int role) {
  if (xyzz_is_inf(a)) return b;                  // (uniform inside the quad: its lanes hold the same a, b)
  if (xyzz_is_inf(b)) return a;
  // step 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
  const Fq t1 = fe_mulu(fe_sel4(role, a.X, b.X, a.Y, b.Y), fe_sel4(role, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ));
  const Fq U1 = fe_quad_bcast<0>(t1), U2 = fe_quad_bcast<1>(t1), S1 = fe_quad_bcast<2>(t1), S2 = fe_quad_bcast<3>(t1);
  const Fq P = fe_normu(fe_subb<FqP, 2, 1>(U2, U1)), R = fe_normu(fe_subb<FqP, 2, 1>(S2, S1));
  if (fe_maybe_zero(P) && fe_is_zero(P)) {
    if (fe_is_zero(R)) return xyzz_dbl(a);           // (inlined: an out-of-line copy gives the kernel a scratch segment, measured +15 us per launch)
    return xyzz_inf();
  }
  // step 2: PP = P P | RR = R R | Z12 = ZZ1 ZZ2 | Z123 = ZZZ1 ZZZ2   (the last two stay with their lanes)
  const Fq t2 = fe_mulu(fe_sel4(role, P, R, a.ZZ, a.ZZZ), fe_sel4(role, P, R, b.ZZ, b.ZZZ));
  const Fq PP = fe_quad_bcast<0>(t2), RR = fe_quad_bcast<1>(t2);
  // step 3: PPP = P PP | Q = U1 PP | ZZ3 = Z12 PP | (lane 3: a spare P PP)
  const Fq t3 = fe_mulu(fe_sel4(role, P, U1, t2, P), PP);
  const Fq PPP = fe_quad_bcast<0>(t3), Q = fe_quad_bcast<1>(t3);
  XYZZ r;
  r.ZZ = fe_quad_bcast<2>(t3);
  r.X = fe_normu(fe_subb<FqP, 4, 3>(RR, fe_add_lazy(fe_add_lazy(PPP, Q), Q)));
  // step 4: T1 = R (Q - X3) | T2 = S1 PPP | (lane 2: a spare) | ZZZ3 = Z123 PPP
  const Fq t4 = fe_mulu(fe_sel4(role, R, S1, S1, t2), fe_sel2(role == 0, PPP, fe_subb<FqP, 6, 1>(Q, r.X)));
  r.Y = fe_normu(fe_subb<FqP, 2, 1>(fe_quad_bcast<0>(t4), fe_quad_bcast<1>(t4)));
  r.ZZZ = fe_quad_bcast<3>(t4);
  return r;
}
template <int K> __device__ __forceinline__ XYZZ xyzz_quad_bcast(const XYZZ& p) {
  XYZZ r; r.X = fe_quad_bcast<K>(p.X); r.Y = fe_quad_bcast<K>(p.Y); r.ZZ = fe_quad_bcast<K>(p.ZZ); r.ZZZ = fe_quad_bcast<K>(p.ZZZ); return r;
}
// the two operands of a butterfly step at distance d, in the same order in both groups (lower group's value first)
__device__ __forceinline__ void xyzz_butterfly_operands(const XYZZ& v, int d, int lane, XYZZ& a, XYZZ& b) {
  const XYZZ o = xyzz_shfl_xor(v, d);
  const bool up = (lane & d) != 0;
  a.X = fe_sel2(up, v.X, o.X); a.Y = fe_sel2(up, v.Y, o.Y); a.ZZ = fe_sel2(up, v.ZZ, o.ZZ); a.ZZZ = fe_sel2(up, v.ZZZ, o.ZZZ);
  b.X = fe_sel2(up, o.X, v.X); b.Y = fe_sel2(up, o.Y, v.Y); b.ZZ = fe_sel2(up, o.ZZ, v.ZZ); b.ZZZ = fe_sel2(up, o.ZZZ, v.ZZZ);
}
// Sum over a block of 64 .. 256 lanes, returned in lanes 0..3 of wave 0 (with one wave: in every lane).  sm: 4 x 128 B of LDS.
// 7 quad additions per wave — a quad first adds its own four points (p0 + p1, p2 + p3, then the two sums: every lane gets all
// four by DPP), then quads pair up by butterfly exchange — and 1 or 2 more across the waves (quad q of wave 0 adds the totals of
// waves 2q and 2q + 1 from LDS, then the two quads add each other's result).  ONE loop with ONE inlined addition: these kernels
// run each instruction once, so their size is instruction-fetch time.
__device__ __forceinline__ XYZZ block_sum_quad(const XYZZ& v, uint32_t (*sm)[32]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6), role = lane & 3;
  const int nsteps = 7 + (nw > 1 ? 1 : 0) + (nw > 2 ? 1 : 0);
  XYZZ keep = v, cur = v;
#pragma unroll 1
  for (int step = 0; step < nsteps; step++) {
    XYZZ a, b;
    if (step == 0) { a = xyzz_quad_bcast<0>(keep); b = xyzz_quad_bcast<1>(keep); }
    else if (step == 1) { a = xyzz_quad_bcast<2>(keep); b = xyzz_quad_bcast<3>(keep); keep = cur; }
    else if (step == 2) { a = keep; b = cur; }
    else if (step == 7) {
      if (lane == 0) xyzz_store(sm[wv], cur);
      __syncthreads();
      const int q = lane >> 2;
      a = (2 * q < nw) ? xyzz_load(sm[(2 * q) & 3]) : xyzz_inf(); b = (2 * q + 1 < nw) ? xyzz_load(sm[(2 * q + 1) & 3]) : xyzz_inf();
    }
    else xyzz_butterfly_operands(cur, step == 8 ? 4 : (4 << (step - 3)), lane, a, b);
    cur = xyzz_add_quad(a, b, role);
  }
  return cur;
}
__device__ __forceinline__ XYZZ wave_sum_all(const XYZZ& v) { return block_sum_quad(v, nullptr); }    // blockDim.x == 64 only

}  // namespace sbn
