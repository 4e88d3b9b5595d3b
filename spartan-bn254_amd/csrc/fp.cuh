// fp.cuh — BN254 Fq / Fr arithmetic for gfx950 (CDNA4): 9 x 29-bit limbs, signed, lazy; Montgomery form R = 2^261.
//
// Stands in for what the reference gets from ark-ff's Fp256 Montgomery backend through `Scalar(Fr)` (reference
// src/scalar.rs:15,114-176) and the Fq coordinates of `G1Affine` (reference src/group.rs:171-175).  Designed from the
// instruction costs MEASURED on MI355X (tools/micro/ibench.hip, cycles per wave-instruction per SIMD with >= 2 waves):
//      v_add_u32 / v_sub_u32 / v_and_b32 / shifts by a constant (plain VOP2)                      2.3
//      v_mad_u64_u32 / v_mad_i64_i32 / v_mul_lo_u32 / every VOP3 / every carry-writing add (v_addc_co_u32 ...)   4.2
// So a carry capture costs as much as a multiply.  The round-1 layer (8 x 32-bit limbs, tools/micro/legacy/fp32.cuh) spends
// 128 multiplies + 8 quotient digits but also 105 carry captures and 50 carry-chained adds per product, and 24 carry-chained
// instructions per addition.  With 29-bit limbs a 64-bit column holds 18 limb products with room to spare: no capture exists,
// a product is 162 multiplies + 9 quotient digits + 27 shifts/adds (measured 1.61e11 /s against 1.29e11 /s, squaring 1.97e11 /s:
// tools/micro/fp29test.hip), and an addition is 9 plain v_add_u32 with no carry chain at all.
//
// Representation.  value = sum_k (int32)v[k] * 2^(29 k).  Limbs are SIGNED and values are LAZY:
//   * "normalised" (what fe_mul / fe_sqr / fe_norm return): v[0..7] in [0, 2^29), v[8] a small signed top limb;
//   * the value is only known modulo p; a product of inputs |a b| < 169 p^2 returns a value in (-p, 2p), and in
//     (-0.1p, 1.1p) when |a|, |b| < 2p (R / p = 2^7.4 = 169: the reduction is strong because R is wide).
//   * operand limits of fe_mul (64-bit signed columns: 9 |a_k| |b_j| + 9 * 2^58 < 2^63): |a_k| * |b_j| <= 2^59.3, e.g. both
//     normalised, or one a difference of two normalised values (|.| < 2^29) — NOT two sums (2^30 each); fe_sqr: |a_k| < 2^29.
//   * fe_add / fe_sub / fe_dbl / fe_neg normalise (always safe as operands); the *_lazy forms are single limb-wise instructions
//     for code that has checked the limits above (g1.cuh's mixed addition, the sumcheck rounds).
//   * memory format: 32 bytes, the value as a non-negative integer < 2^256 (8 x 32-bit words, little-endian), Montgomery domain
//     2^261.  fe_store writes the canonical representative in [0, p); the *_lazy stores write any non-negative representative
//     (callers bound it) and exist for the hot kernels' own intermediate tables.  Zero is stored as zero by both.
// Equality and zero tests are modulo p (they canonicalise), except fe_is_zero_limbs, the exact test of the all-zero
// representation that marks the point at infinity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbn {

constexpr int NL = 9;                         // limbs per element
constexpr uint32_t LMASK = (1u << 29) - 1;

struct FqP {
  // the modulus as 8 x 32-bit words (memory format, canonical test)
  static constexpr uint32_t P0 = 0xd87cfd47u, P1 = 0x3c208c16u, P2 = 0x6871ca8du, P3 = 0x97816a91u,
                            P4 = 0x8181585du, P5 = 0xb85045b6u, P6 = 0xe131a029u, P7 = 0x30644e72u;
  static constexpr uint32_t NINV29 = 0x04866389u;            // -p^-1 mod 2^29
  static constexpr uint32_t PINV29 = (0u - NINV29) & LMASK;  //  p^-1 mod 2^29
  static constexpr uint32_t P29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t ONE29[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};   // 2^261 mod p
  static constexpr uint32_t R2_29[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u, 0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};   // 2^522 mod p
  static constexpr uint32_t C256_29[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u}; // 2^256 mod p: internal -> ark-ff's Montgomery form
  static constexpr uint32_t CIN_29[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};  // 2^266 mod p: ark-ff's Montgomery form -> internal
  static constexpr uint32_t ONE[8] = {0x157ccc21u, 0x4e8384ebu, 0x0ce148c3u, 0xfb90a602u, 0x819caa36u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};                  // 2^261 mod p as 32-bit words (host code)
};
struct FrP {
  static constexpr uint32_t P0 = 0xf0000001u, P1 = 0x43e1f593u, P2 = 0x79b97091u, P3 = 0x2833e848u,
                            P4 = 0x8181585du, P5 = 0xb85045b6u, P6 = 0xe131a029u, P7 = 0x30644e72u;
  static constexpr uint32_t NINV29 = 0x0fffffffu;
  static constexpr uint32_t PINV29 = (0u - NINV29) & LMASK;
  static constexpr uint32_t P29[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t ONE29[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
  static constexpr uint32_t R2_29[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
  static constexpr uint32_t C256_29[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
  static constexpr uint32_t CIN_29[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
  static constexpr uint32_t ONE[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
};

template <class M>
struct Fe {
  uint32_t v[NL];
};
using Fq = Fe<FqP>;
using Fr = Fe<FrP>;

// 32-bit words of the modulus (memory format)
template <class M> __host__ __device__ __forceinline__ constexpr uint32_t modlimb(int i) {
  return i == 0 ? M::P0 : i == 1 ? M::P1 : i == 2 ? M::P2 : i == 3 ? M::P3 : i == 4 ? M::P4 : i == 5 ? M::P5 : i == 6 ? M::P6 : M::P7;
}
// 29-bit limbs of the modulus / of a constant table, as immediates (a switch the compiler folds after unrolling)
#define SBN_C9(T, k) ((k) == 0 ? T[0] : (k) == 1 ? T[1] : (k) == 2 ? T[2] : (k) == 3 ? T[3] : (k) == 4 ? T[4] : (k) == 5 ? T[5] : (k) == 6 ? T[6] : (k) == 7 ? T[7] : T[8])
template <class M> __device__ __forceinline__ constexpr uint32_t p29(int k) { return SBN_C9(M::P29, k); }

template <class M> __device__ __forceinline__ Fe<M> fe_zero() { Fe<M> r; for (int i = 0; i < NL; i++) r.v[i] = 0; return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_one() { Fe<M> r; for (int i = 0; i < NL; i++) r.v[i] = SBN_C9(M::ONE29, i); return r; }   // Montgomery one
template <class M> __device__ __forceinline__ Fe<M> fe_const_r2() { Fe<M> r; for (int i = 0; i < NL; i++) r.v[i] = SBN_C9(M::R2_29, i); return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_const_c256() { Fe<M> r; for (int i = 0; i < NL; i++) r.v[i] = SBN_C9(M::C256_29, i); return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_const_cin() { Fe<M> r; for (int i = 0; i < NL; i++) r.v[i] = SBN_C9(M::CIN_29, i); return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_small(uint32_t x) { Fe<M> r = fe_zero<M>(); r.v[0] = x & LMASK; r.v[1] = x >> 29; return r; }           // the plain integer x

// the exact all-zero representation (infinity marker, empty accumulators): NOT a test modulo p
template <class M> __device__ __forceinline__ bool fe_is_zero_limbs(const Fe<M>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) o |= a.v[i];
  return o == 0;
}

// ---- limb-wise (lazy) operations: one plain VOP2 instruction per limb, no carries ----------------------------------
template <class M> __device__ __forceinline__ Fe<M> fe_add_lazy(const Fe<M>& a, const Fe<M>& b) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
  return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_sub_lazy(const Fe<M>& a, const Fe<M>& b) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] - b.v[i];
  return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_neg_lazy(const Fe<M>& a) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = 0u - a.v[i];
  return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_dbl_lazy(const Fe<M>& a) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = a.v[i] << 1;
  return r; }
// signed carry propagation: v[0..7] into [0, 2^29), the top limb takes what is left (value unchanged)
template <class M> __device__ __forceinline__ Fe<M> fe_norm(Fe<M> a) {
#pragma unroll
  for (int i = 0; i < NL - 1; i++) { const int32_t c = (int32_t)a.v[i] >> 29; a.v[i] &= LMASK; a.v[i + 1] += (uint32_t)c; }
  return a;
}
// the safe forms: normalised results (the value still grows: long accumulations call fe_reduce now and then)
template <class M> __device__ __forceinline__ Fe<M> fe_add(const Fe<M>& a, const Fe<M>& b) { return fe_norm(fe_add_lazy(a, b)); }
template <class M> __device__ __forceinline__ Fe<M> fe_sub(const Fe<M>& a, const Fe<M>& b) { return fe_norm(fe_sub_lazy(a, b)); }
template <class M> __device__ __forceinline__ Fe<M> fe_neg(const Fe<M>& a) { return fe_norm(fe_neg_lazy(a)); }
template <class M> __device__ __forceinline__ Fe<M> fe_dbl(const Fe<M>& a) { return fe_norm(fe_dbl_lazy(a)); }

// ---- the multiplier -------------------------------------------------------------------------------------------------
// Word-serial Montgomery product a*b*2^-261 over a sliding window of nine signed 64-bit columns.  Round i: column k += a[k]*b[i];
// m = the 29-bit digit that clears the lowest column; column k += m*p[k]; the cleared column's upper part moves up and the
// window slides.  Written in C on purpose: hipcc turns every line into the one instruction it should be (v_mad_i64_i32 /
// v_mad_u64_u32 with the column as the 64-bit addend, v_mul_lo_u32 + v_and_b32 for m, v_ashrrev_i64 + v_lshl_add_u64 for the
// carry) — checked in the ISA (tools/isa_stats.py) — and can schedule freely around it.
template <class M, bool SQR> __device__ __forceinline__ Fe<M> fe_mul_impl(const Fe<M>& a, const Fe<M>& b) {
  int64_t c[NL];
  int32_t a2[NL];
  if (SQR) {
#pragma unroll
    for (int k = 0; k < NL; k++) a2[k] = (int32_t)(a.v[k] << 1);
  }
#pragma unroll
  for (int i = 0; i < NL; i++) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
      int64_t t = 0; bool has = true;
      if (!SQR) t = (int64_t)(int32_t)a.v[k] * (int64_t)(int32_t)b.v[i];
      else if (k == i) t = (int64_t)(int32_t)a.v[k] * (int64_t)(int32_t)a.v[i];          // a_i^2 lands in column 2i
      else if (k > i) t = (int64_t)a2[k] * (int64_t)(int32_t)a.v[i];                       // 2 a_k a_i, k > i, lands in column i + k
      else has = false;
      if (i == 0 || k == NL - 1) c[k] = has ? t : 0;                                        // the window's top column is fresh every round
      else if (has) c[k] += t;
    }
    const uint32_t m = ((uint32_t)c[0] * M::NINV29) & LMASK;
#pragma unroll
    for (int k = 0; k < NL; k++) c[k] += (int64_t)((uint64_t)m * (uint64_t)p29<M>(k));
    c[1] += c[0] >> 29;                                                                      // exact: the low 29 bits are zero now
#pragma unroll
    for (int k = 0; k < NL - 1; k++) c[k] = c[k + 1];
  }
  Fe<M> r;
#pragma unroll
  for (int k = 0; k < NL - 1; k++) {
    r.v[k] = (uint32_t)c[k] & LMASK;
    const int64_t carry = c[k] >> 29;
    if (k < NL - 2) c[k + 1] += carry; else r.v[NL - 1] = (uint32_t)carry;
  }
  return r;
}
template <class M> __device__ __forceinline__ Fe<M> fe_mul(const Fe<M>& a, const Fe<M>& b) { return fe_mul_impl<M, false>(a, b); }
template <class M> __device__ __forceinline__ Fe<M> fe_sqr(const Fe<M>& a) { return fe_mul_impl<M, true>(a, a); }

// limb k of K * p with the limbs carried into range (K * P29[k] itself would overflow the signed limb range from K = 4 on)
template <class M, int K> __device__ __forceinline__ constexpr uint32_t kp29(int k) {
  uint64_t carry = 0, t = 0;
  for (int i = 0; i <= k; i++) { t = (uint64_t)K * SBN_C9(M::P29, i) + carry; carry = t >> 29; }
  return k == NL - 1 ? (uint32_t)t : (uint32_t)(t & LMASK);
}
// ---- the unsigned fast path ---------------------------------------------------------------------------------------------
// Measured: products whose operands may be negative cost ~14 % more (hipcc multiplies unsigned and patches the sign in), and the
// signed column limit is 2^63.  The hot formulas therefore keep every limb NON-NEGATIVE: a difference a - b is taken as
// a + (K p) - b with K p written in an "inflated" form whose limbs 0..7 are each >= J * 2^29 - J (what the subtrahends can
// reach: J normalised values) — so no limb goes negative, no carry is needed, and the value only moves by a multiple of p.
// Unsigned columns hold 9 |a_k| |b_j| + 9 * 2^58 < 2^64: |a_k| * |b_j| < 2^60.6 (e.g. a biased difference, < 2^30.6, times a
// normalised value; or two sums of two normalised values).  fe_squ: limbs below 2^30.  Products of non-negative operands
// return a normalised value in [0, a b / 2^261 + p).
template <class M, int K, int J> __device__ __forceinline__ constexpr uint32_t bias29(int k) {
  return k == NL - 1 ? kp29<M, K>(k) - (uint32_t)J : kp29<M, K>(k) + ((uint32_t)J << 29) - (k > 0 ? (uint32_t)J : 0u);
}
// a - b + K p, limb-wise, all limbs non-negative.  b: the sum of up to J normalised values, below (K - 0.001) p in value.
template <class M, int K, int J> __device__ __forceinline__ Fe<M> fe_subb(const Fe<M>& a, const Fe<M>& b) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = (a.v[i] - b.v[i]) + bias29<M, K, J>(i);
  return r; }
// K p - b  (the negation of a normalised value below (K - 0.001) p)
template <class M, int K> __device__ __forceinline__ Fe<M> fe_negb(const Fe<M>& b) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = bias29<M, K, 1>(i) - b.v[i];
  return r; }
// unsigned carry propagation (all limbs non-negative)
template <class M> __device__ __forceinline__ Fe<M> fe_normu(Fe<M> a) {
#pragma unroll
  for (int i = 0; i < NL - 1; i++) { const uint32_t c = a.v[i] >> 29; a.v[i] &= LMASK; a.v[i + 1] += c; }
  return a;
}
template <class M, bool SQR> __device__ __forceinline__ Fe<M> fe_mulu_impl(const Fe<M>& a, const Fe<M>& b) {
  uint64_t c[NL];
  uint32_t a2[NL];
  if (SQR) {
#pragma unroll
    for (int k = 0; k < NL; k++) a2[k] = a.v[k] << 1;
  }
#pragma unroll
  for (int i = 0; i < NL; i++) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
      uint64_t t = 0; bool has = true;
      if (!SQR) t = (uint64_t)a.v[k] * (uint64_t)b.v[i];
      else if (k == i) t = (uint64_t)a.v[k] * (uint64_t)a.v[i];
      else if (k > i) t = (uint64_t)a2[k] * (uint64_t)a.v[i];
      else has = false;
      if (i == 0 || k == NL - 1) c[k] = has ? t : 0;
      else if (has) c[k] += t;
    }
    const uint32_t m = ((uint32_t)c[0] * M::NINV29) & LMASK;
#pragma unroll
    for (int k = 0; k < NL; k++) c[k] += (uint64_t)m * (uint64_t)p29<M>(k);
    c[1] += c[0] >> 29;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) c[k] = c[k + 1];
  }
  Fe<M> r;
#pragma unroll
  for (int k = 0; k < NL - 1; k++) {
    r.v[k] = (uint32_t)c[k] & LMASK;
    const uint64_t carry = c[k] >> 29;
    if (k < NL - 2) c[k + 1] += carry; else r.v[NL - 1] = (uint32_t)carry;
  }
  return r;
}
template <class M> __device__ __forceinline__ Fe<M> fe_mulu(const Fe<M>& a, const Fe<M>& b) { return fe_mulu_impl<M, false>(a, b); }
template <class M> __device__ __forceinline__ Fe<M> fe_squ(const Fe<M>& a) { return fe_mulu_impl<M, true>(a, a); }

// ---- sums of products with ONE reduction ---------------------------------------------------------------------------------
// The 17 columns of a schoolbook product of two normalised, non-negative operands (limbs below 2^29: every limb product is below
// 2^58, a column takes at most 9 of them), kept in 64-bit registers so that SEVERAL products can be added up before the Montgomery
// reduction runs once over the sum: 81 multiplier instructions per extra product instead of 171 (the reduction's 81 + 9 are shared).
// Capacity: 6 products between carry passes / before the reduction (54 * 2^58 + the reduction's own 9 * 2^58 + carries < 2^64).
constexpr int NC = 2 * NL - 1;
struct Cols { uint64_t c[NC]; };
__device__ __forceinline__ void cols_zero(Cols& s) {
#pragma unroll
  for (int k = 0; k < NC; k++) s.c[k] = 0;
}
// s += a (x) b   (both normalised, all limbs non-negative)
template <class M> __device__ __forceinline__ void cols_mac(Cols& s, const Fe<M>& a, const Fe<M>& b) {
#pragma unroll
  for (int i = 0; i < NL; i++) {
#pragma unroll
    for (int j = 0; j < NL; j++) s.c[i + j] += (uint64_t)a.v[i] * (uint64_t)b.v[j];
  }
}
// s += a (x) b with ONE lazy operand: a normalised (limbs below 2^29), b non-negative with limbs below 2^30.6 (a biased difference,
// fe_subb / fe_negb).  Capacity: two such products before the reduction (2 * 9 * 2^59.6 + 9 * 2^58 < 2^64).
template <class M> __device__ __forceinline__ void cols_mac_lazy(Cols& s, const Fe<M>& a, const Fe<M>& b) { cols_mac<M>(s, a, b); }
// carries moved up (value unchanged): columns 0..15 below 2^29 afterwards, the top column takes what is left
__device__ __forceinline__ void cols_carry(Cols& s) {
#pragma unroll
  for (int k = 0; k < NC - 1; k++) { s.c[k + 1] += s.c[k] >> 29; s.c[k] &= (uint64_t)LMASK; }
}
// the sum * 2^-261 mod p as a normalised value in [0, sum / 2^261 + p): nine quotient digits clear the low nine columns
template <class M> __device__ __forceinline__ Fe<M> cols_reduce(Cols s) {
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const uint32_t m = ((uint32_t)s.c[i] * M::NINV29) & LMASK;
#pragma unroll
    for (int k = 0; k < NL; k++) s.c[i + k] += (uint64_t)m * (uint64_t)p29<M>(k);
    s.c[i + 1] += s.c[i] >> 29;                          // exact: the low 29 bits are zero now
  }
  Fe<M> r;
#pragma unroll
  for (int k = 0; k < NL - 1; k++) {
    r.v[k] = (uint32_t)s.c[NL + k] & LMASK;
    const uint64_t carry = s.c[NL + k] >> 29;
    if (k < NL - 2) s.c[NL + k + 1] += carry; else r.v[NL - 1] = (uint32_t)carry;
  }
  return r;
}

// the value into (-0.1p, 1.1p) (for |a| < 13p), normalised: a Montgomery product with one
template <class M> __device__ __forceinline__ Fe<M> fe_reduce(const Fe<M>& a) { return fe_mul<M>(a, fe_one<M>()); }

// x in (-p, 2p), normalised  ->  the representative in [0, p), normalised
template <class M> __device__ __forceinline__ Fe<M> fe_canon_small(Fe<M> x) {
  const uint32_t neg = (uint32_t)((int32_t)x.v[NL - 1] >> 31);          // normalised: value < 0 <=> top limb < 0
#pragma unroll
  for (int k = 0; k < NL; k++) x.v[k] += p29<M>(k) & neg;
  x = fe_norm(x);                                                        // [0, 2p)
  Fe<M> y;
#pragma unroll
  for (int k = 0; k < NL; k++) y.v[k] = x.v[k] - p29<M>(k);
  y = fe_norm(y);                                                        // x - p in [-p, p)
  const uint32_t keep = (uint32_t)((int32_t)y.v[NL - 1] >> 31);          // x < p: keep x
#pragma unroll
  for (int k = 0; k < NL; k++) x.v[k] = (x.v[k] & keep) | (y.v[k] & ~keep);
  return x;
}
// any lazy value (|a| < 13p, limbs within fe_mul's operand limits) -> the representative in [0, p), normalised
template <class M> __device__ __forceinline__ Fe<M> fe_canon(const Fe<M>& a) { return fe_canon_small<M>(fe_reduce<M>(a)); }

template <class M> __device__ __forceinline__ bool fe_is_zero(const Fe<M>& a) { return fe_is_zero_limbs(fe_canon<M>(a)); }      // modulo p
template <class M> __device__ __forceinline__ bool fe_eq(const Fe<M>& a, const Fe<M>& b) { return fe_is_zero<M>(fe_sub_lazy<M>(fe_norm(a), fe_norm(b))); }
// Cheap filter for "d = 0 (mod p)" on a lazy difference with |d| < 8p: such a d is k*p with |k| <= 8, so d * p^-1 mod 2^29 is
// k mod 2^29.  One quotient digit instead of a canonicalisation; false positives (9 / 2^29 per call) go to the exact test.
template <class M> __device__ __forceinline__ bool fe_maybe_zero(const Fe<M>& d) {
  const uint32_t k = (d.v[0] * M::PINV29 + 8u) & LMASK;
  return k <= 16u;
}

// canonical integer (limbs of a value < p, e.g. just unpacked) <-> Montgomery form
template <class M> __device__ __forceinline__ Fe<M> fe_to_mont(const Fe<M>& a) { return fe_mul<M>(a, fe_const_r2<M>()); }
template <class M> __device__ __forceinline__ Fe<M> fe_from_mont(const Fe<M>& a) { return fe_canon_small<M>(fe_mul<M>(a, fe_small<M>(1u))); }     // canonical plain integer
// ark-ff's in-memory Montgomery limbs (R = 2^256) <-> internal form (R = 2^261)
template <class M> __device__ __forceinline__ Fe<M> fe_from_ark_mont(const Fe<M>& a) { return fe_mul<M>(a, fe_const_cin<M>()); }
template <class M> __device__ __forceinline__ Fe<M> fe_ark_mont_to_plain(const Fe<M>& a) { return fe_canon_small<M>(fe_mul<M>(a, fe_small<M>(32u))); }   // x*2^256 * 2^5 * 2^-261 = x

// a^(p-2): Fermat inversion, only used once per output point (never per input point).  Input normalised.
template <class M> __device__ __noinline__ Fe<M> fe_inv(const Fe<M>& a) {
  uint32_t e[8];
  for (int i = 0; i < 8; i++) e[i] = modlimb<M>(i);
  e[0] -= 2;  // p0 >= 2 for both moduli, no borrow
  Fe<M> acc = fe_one<M>();
  for (int i = 255; i >= 0; i--) {
    acc = fe_sqr<M>(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul<M>(acc, a);
  }
  return acc;
}

// value < modulus?  (8 little-endian 32-bit words; what Scalar::from_bytes checks, scalar.rs:87-95)
template <class M> __device__ __forceinline__ bool fe_is_canonical(const uint32_t k[8]) {
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)k[i] - modlimb<M>(i) - borrow; borrow = d >> 63; }
  return borrow != 0;        // k - p borrowed  <=>  k < p
}

// ---- memory format: 32 bytes = 8 x 32-bit words of a non-negative integer < 2^256 ------------------------------------
template <class M> __device__ __forceinline__ Fe<M> fe_unpack(const uint32_t w[8]) {
  Fe<M> r;
#pragma unroll
  for (int k = 0; k < NL; k++) {
    const int bit = 29 * k, i = bit >> 5, sh = bit & 31;
    uint32_t x = w[i] >> sh;
    if (sh > 3 && i + 1 < 8) x |= w[i + 1] << (32 - sh);        // the limb straddles two words
    r.v[k] = x & LMASK;
  }
  return r;
}
// a: normalised, value in [0, 2^256)
template <class M> __device__ __forceinline__ void fe_pack(const Fe<M>& a, uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int lo = (32 * i) / 29, sh = 32 * i - 29 * lo;
    uint32_t x = a.v[lo] >> sh;
    x |= a.v[lo + 1] << (29 - sh);
    if (58 - sh < 32 && lo + 2 < NL) x |= a.v[lo + 2] << (58 - sh);
    w[i] = x;
  }
}
// normalised x with value in (-K p, 2^256 - K p): add K*p when negative, so that the stored integer is non-negative; zero stays zero
template <class M, int K> __device__ __forceinline__ Fe<M> fe_fix_nonneg(Fe<M> x) {
  const uint32_t neg = (uint32_t)((int32_t)x.v[NL - 1] >> 31);
#pragma unroll
  for (int k = 0; k < NL; k++) x.v[k] += kp29<M, K>(k) & neg;
  return fe_norm(x);
}

// Intermediate tables (sumcheck tables, partial sums): normalised x with value in (-2p, 4.5p) -> a non-negative representative in
// [0, 2.5p): + p when negative (+ 2p below -p), - 2p from 2p up.  The decisions read the top limb only: p = (P8 + f) * 2^232.
template <class M> __device__ __forceinline__ Fe<M> fe_fix_tab(Fe<M> x) {
  constexpr int32_t P8 = (int32_t)M::P29[8];
  const int32_t top = (int32_t)x.v[NL - 1];
  const uint32_t m1 = (uint32_t)(top >> 31);                         // value < 0
  const uint32_t m2 = (uint32_t)((top + P8) >> 31);                  // top <= -P8 - 1: value < -P8 * 2^232 ~ -p
  const uint32_t m3 = (uint32_t)((2 * P8 + 1 - top) >> 31);          // top >= 2 P8 + 2: value > 2p
#pragma unroll
  for (int k = 0; k < NL; k++) x.v[k] += (p29<M>(k) & m1) + (p29<M>(k) & m2) - ((2u * p29<M>(k)) & m3);
  return fe_norm(x);
}
template <class M> __device__ __forceinline__ Fe<M> fe_from_u64(unsigned long long x) {      // the plain integer x
  Fe<M> r = fe_zero<M>(); r.v[0] = (uint32_t)x & LMASK; r.v[1] = (uint32_t)(x >> 29) & LMASK; r.v[2] = (uint32_t)(x >> 58); return r;
}

typedef uint32_t sbn_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) sbn_u32x4 sbn_g_u32x4;
// 16-byte vector load/store of a field element (AoS, 32 B, 16 B aligned)
template <class M> __device__ __forceinline__ Fe<M> fe_load(const void* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return fe_unpack<M>(w);
}
template <class M> __device__ __forceinline__ void fe_store_packed(void* p, const Fe<M>& a /* normalised, in [0, 2^256) */) {
  uint32_t w[8]; fe_pack<M>(a, w);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
template <class M> __device__ __forceinline__ void fe_store(void* p, const Fe<M>& a) { fe_store_packed<M>(p, fe_canon<M>(a)); }      // canonical: [0, p)
template <class M> __device__ __forceinline__ void fe_store_tab(void* p, const Fe<M>& a /* lazy, value in (-2p, 4.5p) */) { fe_store_packed<M>(p, fe_fix_tab<M>(fe_norm(a))); }
// The same on a pointer KNOWN to be global memory.  Pointers that reach a kernel inside a by-value argument struct, or that are
// loaded from memory (pointer tables), are generic to the compiler: it emits flat_load / flat_store, and a flat access counts on
// both vmcnt and lgkmcnt and returns out of order, so every wait degenerates to `s_waitcnt vmcnt(0) lgkmcnt(0)` and no load can
// stay in flight across a use (measured on the fused sumcheck round: the prefetch of the next table was drained by the very next
// wait).  The explicit address-space-1 access types make the instruction itself global_load_dwordx4 / global_store_dwordx4.
template <class M> __device__ __forceinline__ Fe<M> fe_gload(const void* p) {
  const sbn_g_u32x4* q = (const sbn_g_u32x4*)p;
  const sbn_u32x4 lo = q[0], hi = q[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return fe_unpack<M>(w);
}
template <class M> __device__ __forceinline__ void fe_gstore_packed(void* p, const Fe<M>& a) {
  uint32_t w[8]; fe_pack<M>(a, w);
  sbn_g_u32x4* q = (sbn_g_u32x4*)p;
  sbn_u32x4 lo, hi;
  lo.x = w[0]; lo.y = w[1]; lo.z = w[2]; lo.w = w[3]; hi.x = w[4]; hi.y = w[5]; hi.z = w[6]; hi.w = w[7];
  q[0] = lo; q[1] = hi;
}
template <class M> __device__ __forceinline__ void fe_gstore(void* p, const Fe<M>& a) { fe_gstore_packed<M>(p, fe_canon<M>(a)); }
template <class M> __device__ __forceinline__ void fe_gstore_tab(void* p, const Fe<M>& a) { fe_gstore_packed<M>(p, fe_fix_tab<M>(fe_norm(a))); }

// wave64 exchange of a whole element
template <class M> __device__ __forceinline__ Fe<M> fe_shfl_down(const Fe<M>& a, int delta) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = __shfl_down(a.v[i], delta, 64);
  return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_shfl_xor(const Fe<M>& a, int mask) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = __shfl_xor(a.v[i], mask, 64);
  return r; }

// ---- quad helpers (latency-bound sums, g1.cuh: xyzz_add_quad): lanes 4q .. 4q+3 work on one addition ----
// value of lane K of the caller's quad, in all four lanes (DPP quad_perm: no LDS, no waitcnt)
template <int K, class M> __device__ __forceinline__ Fe<M> fe_quad_bcast(const Fe<M>& a) { Fe<M> r;
#pragma unroll
  // (update_dpp with old = src; __builtin_amdgcn_mov_dpp with bound_ctrl gave wrong values in two lanes of every quad once the
  //  compiler folded it into the consuming instruction — measured with tools/micro/g1test.hip)
  for (int i = 0; i < NL; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)a.v[i], (int)a.v[i], K * 0x55, 0xf, 0xf, false);
  return r; }
// role-indexed operand: role 0..3 picks a0..a3 (limb-wise selects; the four values need not be normalised alike)
template <class M> __device__ __forceinline__ Fe<M> fe_sel4(int role, const Fe<M>& a0, const Fe<M>& a1, const Fe<M>& a2, const Fe<M>& a3) { Fe<M> r;
  const bool hi = (role & 2) != 0, odd = (role & 1) != 0;
#pragma unroll
  for (int i = 0; i < NL; i++) { const uint32_t lo2 = odd ? a1.v[i] : a0.v[i], hi2 = odd ? a3.v[i] : a2.v[i]; r.v[i] = hi ? hi2 : lo2; }
  return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_sel2(bool second, const Fe<M>& a0, const Fe<M>& a1) { Fe<M> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = second ? a1.v[i] : a0.v[i];
  return r; }

}  // namespace sbn
