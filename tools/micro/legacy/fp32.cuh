// LEGACY (round 1): the 8 x 32-bit-limb field layer, kept ONLY as the reference the micro-benchmarks compare the shipped 9 x 29-bit layer
// (spartan-bn254_amd/csrc/fp.cuh) against.  Nothing under spartan-bn254_amd/ includes this file.
// fp.cuh — BN254 Fq / Fr arithmetic for gfx950 (CDNA4), 8 x 32-bit limbs, Montgomery form R = 2^256.
//
// Stands in for what the reference gets from ark-ff's Fp256 Montgomery backend through
// `Scalar(Fr)` (reference src/scalar.rs:15,114-176) and the Fq coordinates of `G1Affine`
// (reference src/group.rs:171-175).  Re-derived for the CDNA4 VALU, not translated:
//
//  * The only wide multiplier on the vector ALU is v_mad_u64_u32 (32x32+64 -> 64), measured
//    half-rate on MI355X (tools/micro/ubench.hip: ~4.8 cyc/wave-instr vs 2.3 for v_add).  A 256-bit
//    Montgomery product needs 128 of them + 8 v_mul_lo_u32; everything else is overhead to minimise.
//  * v_mad_u64_u32 has a carry-OUT (SGPR pair) but no carry-in, so a CIOS written in C costs
//    one 64-bit add + register-pair shuffles per product (hipcc: 1583 cyc/modmul/wave).  Here every limb
//    position owns a 64-bit accumulator pair (limbs k,k+1; even and odd positions never alias) that
//    products are added into directly, and the 1-bit overflow of each mad is captured into a per-position
//    counter by one full-rate v_addc_co_u32.  Counters are folded in once per row shift / at the end.
//  * p < 2^254 (two spare bits) => the top accumulator of a row can never overflow; no capture there.
//
// All values are kept fully reduced in [0, p).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbn32 {

struct FqP {
  static constexpr uint32_t P0 = 0xd87cfd47u, P1 = 0x3c208c16u, P2 = 0x6871ca8du, P3 = 0x97816a91u,
                            P4 = 0x8181585du, P5 = 0xb85045b6u, P6 = 0xe131a029u, P7 = 0x30644e72u;
  static constexpr uint32_t NINV = 0xe4866389u;  // -p^-1 mod 2^32
  // R mod p (Montgomery one) and R^2 mod p
  static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct FrP {
  static constexpr uint32_t P0 = 0xf0000001u, P1 = 0x43e1f593u, P2 = 0x79b97091u, P3 = 0x2833e848u,
                            P4 = 0x8181585du, P5 = 0xb85045b6u, P6 = 0xe131a029u, P7 = 0x30644e72u;
  static constexpr uint32_t NINV = 0xefffffffu;
  static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class M>
struct alignas(16) Fe {
  uint32_t v[8];
};
using Fq = Fe<FqP>;
using Fr = Fe<FrP>;

template <class M> __device__ __forceinline__ constexpr uint32_t modlimb(int i) {
  return i == 0 ? M::P0 : i == 1 ? M::P1 : i == 2 ? M::P2 : i == 3 ? M::P3 : i == 4 ? M::P4 : i == 5 ? M::P5 : i == 6 ? M::P6 : M::P7;
}

template <class M> __device__ __forceinline__ Fe<M> fe_zero() { Fe<M> r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
template <class M> __device__ __forceinline__ Fe<M> fe_one() { Fe<M> r; for (int i = 0; i < 8; i++) r.v[i] = M::ONE[i]; return r; }
template <class M> __device__ __forceinline__ bool fe_is_zero(const Fe<M>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
template <class M> __device__ __forceinline__ bool fe_eq(const Fe<M>& a, const Fe<M>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// The carry chains are written as asm: from C, hipcc lowers them to v_lshl_add_u64 plus register-pair
// shuffles (3-4 instructions per limb instead of 1).
// (a carry-chained VOP2 already reads VCC on the constant bus, so the modulus limbs must sit in VGPRs here)
#define SBN_PS(M) "v"(modlimb<M>(0)), "v"(modlimb<M>(1)), "v"(modlimb<M>(2)), "v"(modlimb<M>(3)), \
                  "v"(modlimb<M>(4)), "v"(modlimb<M>(5)), "v"(modlimb<M>(6)), "v"(modlimb<M>(7))

// r = t - p if t >= p else t   (t < 2p < 2^256)
template <class M> __device__ __forceinline__ void fe_cond_sub_p(uint32_t r[8], const uint32_t t[8]) {
  asm("v_subrev_co_u32 %0, vcc, %16, %8\n\tv_subbrev_co_u32 %1, vcc, %17, %9, vcc\n\t"
      "v_subbrev_co_u32 %2, vcc, %18, %10, vcc\n\tv_subbrev_co_u32 %3, vcc, %19, %11, vcc\n\t"
      "v_subbrev_co_u32 %4, vcc, %20, %12, vcc\n\tv_subbrev_co_u32 %5, vcc, %21, %13, vcc\n\t"
      "v_subbrev_co_u32 %6, vcc, %22, %14, vcc\n\tv_subbrev_co_u32 %7, vcc, %23, %15, vcc\n\t"
      "v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\t"
      "v_cndmask_b32 %2, %2, %10, vcc\n\tv_cndmask_b32 %3, %3, %11, vcc\n\t"
      "v_cndmask_b32 %4, %4, %12, vcc\n\tv_cndmask_b32 %5, %5, %13, vcc\n\t"
      "v_cndmask_b32 %6, %6, %14, vcc\n\tv_cndmask_b32 %7, %7, %15, vcc"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]), SBN_PS(M)
      : "vcc");
}

template <class M> __device__ __forceinline__ Fe<M> fe_add(const Fe<M>& a, const Fe<M>& b) {
  uint32_t t[8];
  asm("v_add_co_u32 %0, vcc, %8, %16\n\tv_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\tv_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\tv_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\tv_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  Fe<M> r; fe_cond_sub_p<M>(r.v, t);  // a+b < 2p < 2^255: never carries out of limb 7
  return r;
}
// a - b, plus p under a mask when the subtraction borrowed
template <class M> __device__ __forceinline__ Fe<M> fe_sub(const Fe<M>& a, const Fe<M>& b) {
  uint32_t t[8]; uint32_t mask;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\tv_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\tv_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\tv_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\tv_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_cndmask_b32 %8, 0, -1, vcc"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(mask)
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  uint32_t q[8];
#pragma unroll
  for (int i = 0; i < 8; i++) q[i] = modlimb<M>(i) & mask;
  Fe<M> r;
  asm("v_add_co_u32 %0, vcc, %8, %16\n\tv_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\tv_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\tv_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\tv_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
      : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]),
        "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
      : "vcc");
  return r;
}
template <class M> __device__ __forceinline__ Fe<M> fe_neg(const Fe<M>& a) {
  Fe<M> r = fe_sub<M>(fe_zero<M>(), a);
  return r;
}
template <class M> __device__ __forceinline__ Fe<M> fe_dbl(const Fe<M>& a) { return fe_add<M>(a, a); }

// ---- the multiplier -------------------------------------------------------------------------
// One asm statement per half-row keeps hipcc from padding an s_nop after every single mad
// (it pads one wait state after each asm statement it cannot see into).
//
// SBN_ROW_AB: A[k] += a[k]*b for k = 0..6 with overflow capture into C[k+2];  A[7] = a[7]*b (fresh).
#define SBN_ROW_AB(A, C, a, b)                                                                  \
  asm("v_mad_u64_u32 %0, vcc, %15, %23, %0\n\tv_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"            \
      "v_mad_u64_u32 %1, vcc, %16, %23, %1\n\tv_addc_co_u32 %9, vcc, 0, %9, vcc\n\t"            \
      "v_mad_u64_u32 %2, vcc, %17, %23, %2\n\tv_addc_co_u32 %10, vcc, 0, %10, vcc\n\t"          \
      "v_mad_u64_u32 %3, vcc, %18, %23, %3\n\tv_addc_co_u32 %11, vcc, 0, %11, vcc\n\t"          \
      "v_mad_u64_u32 %4, vcc, %19, %23, %4\n\tv_addc_co_u32 %12, vcc, 0, %12, vcc\n\t"          \
      "v_mad_u64_u32 %5, vcc, %20, %23, %5\n\tv_addc_co_u32 %13, vcc, 0, %13, vcc\n\t"          \
      "v_mad_u64_u32 %6, vcc, %21, %23, %6\n\tv_addc_co_u32 %14, vcc, 0, %14, vcc\n\t"          \
      "v_mad_u64_u32 %7, vcc, %22, %23, 0"                                                       \
      : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "=&v"(A[7]), \
        "+v"(C[2]), "+v"(C[3]), "+v"(C[4]), "+v"(C[5]), "+v"(C[6]), "+v"(C[7]), "+v"(C[8])       \
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b) \
      : "vcc")
// first row: all accumulators fresh, nothing can overflow
#define SBN_ROW_AB0(A, a, b)                                                                    \
  asm("v_mad_u64_u32 %0, vcc, %8, %16, 0\n\tv_mad_u64_u32 %1, vcc, %9, %16, 0\n\t"               \
      "v_mad_u64_u32 %2, vcc, %10, %16, 0\n\tv_mad_u64_u32 %3, vcc, %11, %16, 0\n\t"             \
      "v_mad_u64_u32 %4, vcc, %12, %16, 0\n\tv_mad_u64_u32 %5, vcc, %13, %16, 0\n\t"             \
      "v_mad_u64_u32 %6, vcc, %14, %16, 0\n\tv_mad_u64_u32 %7, vcc, %15, %16, 0"                  \
      : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]), "=&v"(A[4]), "=&v"(A[5]), "=&v"(A[6]), "=&v"(A[7]) \
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b) \
      : "vcc")
// SBN_ROW_MP: A[k] += m*p[k] (p limbs in SGPRs), captures as above; A[7] cannot overflow (a7*b + m*p7 < 2^63);
// then the one-limb right shift: A[0].lo is 0 now, A[0].hi and C[2] fold into A[1], overflow into C[3].
#define SBN_ROW_MP(A, C, m, M)                                                                  \
  asm("v_mad_u64_u32 %0, vcc, %15, %16, %0\n\tv_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"            \
      "v_mad_u64_u32 %1, vcc, %15, %17, %1\n\tv_addc_co_u32 %9, vcc, 0, %9, vcc\n\t"            \
      "v_mad_u64_u32 %2, vcc, %15, %18, %2\n\tv_addc_co_u32 %10, vcc, 0, %10, vcc\n\t"          \
      "v_mad_u64_u32 %3, vcc, %15, %19, %3\n\tv_addc_co_u32 %11, vcc, 0, %11, vcc\n\t"          \
      "v_mad_u64_u32 %4, vcc, %15, %20, %4\n\tv_addc_co_u32 %12, vcc, 0, %12, vcc\n\t"          \
      "v_mad_u64_u32 %5, vcc, %15, %21, %5\n\tv_addc_co_u32 %13, vcc, 0, %13, vcc\n\t"          \
      "v_mad_u64_u32 %6, vcc, %15, %22, %6\n\tv_addc_co_u32 %14, vcc, 0, %14, vcc\n\t"          \
      "v_mad_u64_u32 %7, vcc, %15, %23, %7"                                                      \
      : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), \
        "+v"(C[2]), "+v"(C[3]), "+v"(C[4]), "+v"(C[5]), "+v"(C[6]), "+v"(C[7]), "+v"(C[8])       \
      : "v"(m), "s"(modlimb<M>(0)), "s"(modlimb<M>(1)), "s"(modlimb<M>(2)), "s"(modlimb<M>(3)),    \
        "s"(modlimb<M>(4)), "s"(modlimb<M>(5)), "s"(modlimb<M>(6)), "s"(modlimb<M>(7))             \
      : "vcc")

// Montgomery product a*b*R^-1 mod p, inputs and output in [0,p).
template <class M> __device__ __forceinline__ Fe<M> fe_mul(const Fe<M>& a, const Fe<M>& b) {
  uint64_t A[8];       // A[k] covers limbs k,k+1 of the running total (relative to the current shift)
  uint32_t C[9];       // C[j]: pending overflow count at limb j (j = 2..8)
#pragma unroll
  for (int j = 0; j < 9; j++) C[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t bi = b.v[i];
    if (i == 0) SBN_ROW_AB0(A, a.v, bi); else SBN_ROW_AB(A, C, a.v, bi);
    const uint32_t m = (uint32_t)A[0] * M::NINV;
    SBN_ROW_MP(A, C, m, M);
    // shift right one limb: low limb of A[0] is now 0; its high limb and C[2] fold into A[1]
    uint32_t lo = (uint32_t)A[1], hi = (uint32_t)(A[1] >> 32);
    const uint32_t a0hi = (uint32_t)(A[0] >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc"
        : "+v"(lo), "+v"(hi), "+v"(C[3]) : "v"(a0hi), "v"(C[2]) : "vcc");
    A[0] = ((uint64_t)hi << 32) | lo;
#pragma unroll
    for (int k = 1; k < 7; k++) A[k] = A[k + 1];
#pragma unroll
    for (int j = 2; j < 8; j++) C[j] = C[j + 1];
    C[8] = 0;
  }
  // total = sum_{k<7} A[k] 2^(32k) + sum_{j=2..7} C[j] 2^(32j)  < 2p
  uint32_t lo[8], hi[8];
#pragma unroll
  for (int k = 0; k < 7; k++) { lo[k] = (uint32_t)A[k]; hi[k] = (uint32_t)(A[k] >> 32); }
  uint32_t t[8];
  t[0] = lo[0];
  asm("v_add_co_u32 %0, vcc, %7, %14\n\tv_addc_co_u32 %1, vcc, %8, %15, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %9, %16, vcc\n\tv_addc_co_u32 %3, vcc, %10, %17, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %11, %18, vcc\n\tv_addc_co_u32 %5, vcc, %12, %19, vcc\n\t"
      "v_addc_co_u32 %6, vcc, 0, %13, vcc\n\t"
      "v_add_co_u32 %1, vcc, %1, %20\n\tv_addc_co_u32 %2, vcc, %2, %21, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %22, vcc\n\tv_addc_co_u32 %4, vcc, %4, %23, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %24, vcc\n\tv_addc_co_u32 %6, vcc, %6, %25, vcc"
      : "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
      : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]),
        "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]),
        "v"(C[2]), "v"(C[3]), "v"(C[4]), "v"(C[5]), "v"(C[6]), "v"(C[7])
      : "vcc");
  Fe<M> r; fe_cond_sub_p<M>(r.v, t);
  return r;
}
// The same product with DEFERRED carry captures (fp_rows_gen.inc, generated by tools/gen_fe_mul.py): the 8 mads of a row issue
// back to back, each carry-out into its own SGPR pair, and the captures follow the second row.  Same instruction count, different
// order: no instruction depends on the one before it, so one wave keeps the multiplier busy by itself.
#include "fp_rows_gen.inc"
template <class M> __device__ __forceinline__ Fe<M> fe_mul_deferred(const Fe<M>& a, const Fe<M>& b) {
  uint64_t A[8];
  uint32_t C[9];
#pragma unroll
  for (int j = 0; j < 9; j++) C[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t bi = b.v[i];
    uint64_t S[7];
    if (i == 0) SBN_ROW_AB0(A, a.v, bi); else SBN2_ROW_AB(A, S, a.v, bi);
    const uint32_t m = (uint32_t)A[0] * M::NINV;
    if (i == 0) SBN2_ROW_MP0(A, C, m, M) else SBN2_ROW_MP(A, C, S, m, M)
    uint32_t lo = (uint32_t)A[1], hi = (uint32_t)(A[1] >> 32);
    const uint32_t a0hi = (uint32_t)(A[0] >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc"
        : "+v"(lo), "+v"(hi), "+v"(C[3]) : "v"(a0hi), "v"(C[2]) : "vcc");
    A[0] = ((uint64_t)hi << 32) | lo;
#pragma unroll
    for (int k = 1; k < 7; k++) A[k] = A[k + 1];
#pragma unroll
    for (int j = 2; j < 8; j++) C[j] = C[j + 1];
    C[8] = 0;
  }
  uint32_t lo[8], hi[8];
#pragma unroll
  for (int k = 0; k < 7; k++) { lo[k] = (uint32_t)A[k]; hi[k] = (uint32_t)(A[k] >> 32); }
  uint32_t t[8];
  t[0] = lo[0];
  asm("v_add_co_u32 %0, vcc, %7, %14\n\tv_addc_co_u32 %1, vcc, %8, %15, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %9, %16, vcc\n\tv_addc_co_u32 %3, vcc, %10, %17, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %11, %18, vcc\n\tv_addc_co_u32 %5, vcc, %12, %19, vcc\n\t"
      "v_addc_co_u32 %6, vcc, 0, %13, vcc\n\t"
      "v_add_co_u32 %1, vcc, %1, %20\n\tv_addc_co_u32 %2, vcc, %2, %21, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %22, vcc\n\tv_addc_co_u32 %4, vcc, %4, %23, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %24, vcc\n\tv_addc_co_u32 %6, vcc, %6, %25, vcc"
      : "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
      : "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]),
        "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]),
        "v"(C[2]), "v"(C[3]), "v"(C[4]), "v"(C[5]), "v"(C[6]), "v"(C[7])
      : "vcc");
  Fe<M> r; fe_cond_sub_p<M>(r.v, t);
  return r;
}
template <class M> __device__ __forceinline__ Fe<M> fe_sqr(const Fe<M>& a) { return fe_mul<M>(a, a); }

// canonical integer (8 limbs, < p) <-> Montgomery
template <class M> __device__ __forceinline__ Fe<M> fe_to_mont(const Fe<M>& a) {
  Fe<M> r2; for (int i = 0; i < 8; i++) r2.v[i] = M::R2[i];
  return fe_mul<M>(a, r2);
}
template <class M> __device__ __forceinline__ Fe<M> fe_from_mont(const Fe<M>& a) {
  Fe<M> one; one.v[0] = 1; for (int i = 1; i < 8; i++) one.v[i] = 0;
  return fe_mul<M>(a, one);
}

// a^(p-2): Fermat inversion, only used once per output point (never per input point)
template <class M> __device__ __noinline__ Fe<M> fe_inv(const Fe<M>& a) {
  // exponent p-2, scanned MSB->LSB
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

// value < modulus?  (8 little-endian limbs; what Scalar::from_bytes checks, scalar.rs:87-95)
template <class M> __device__ __forceinline__ bool fe_is_canonical(const uint32_t k[8]) {
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)k[i] - modlimb<M>(i) - borrow; borrow = d >> 63; }
  return borrow != 0;        // k - p borrowed  <=>  k < p
}

// 16-byte vector load/store of a field element (AoS, 32 B, 16 B aligned)
template <class M> __device__ __forceinline__ Fe<M> fe_load(const void* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  Fe<M> r; r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w; r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}
template <class M> __device__ __forceinline__ void fe_store(void* p, const Fe<M>& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// The same on a pointer KNOWN to be global memory.  Pointers that reach a kernel inside a by-value argument struct, or that are
// loaded from memory (pointer tables), are generic to the compiler: it emits flat_load / flat_store, and a flat access counts on
// both vmcnt and lgkmcnt and returns out of order, so every wait degenerates to `s_waitcnt vmcnt(0) lgkmcnt(0)` and no load can
// stay in flight across a use (measured on the fused sumcheck round: the prefetch of the next table was drained by the very next
// wait).  The explicit address-space-1 access types make the instruction itself global_load_dwordx4 / global_store_dwordx4.
typedef uint32_t sbn_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) sbn_u32x4 sbn_g_u32x4;
template <class M> __device__ __forceinline__ Fe<M> fe_gload(const void* p) {
  const sbn_g_u32x4* q = (const sbn_g_u32x4*)p;
  const sbn_u32x4 lo = q[0], hi = q[1];
  Fe<M> r; r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w; r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}
template <class M> __device__ __forceinline__ void fe_gstore(void* p, const Fe<M>& a) {
  sbn_g_u32x4* q = (sbn_g_u32x4*)p;
  sbn_u32x4 lo, hi;
  lo.x = a.v[0]; lo.y = a.v[1]; lo.z = a.v[2]; lo.w = a.v[3]; hi.x = a.v[4]; hi.y = a.v[5]; hi.z = a.v[6]; hi.w = a.v[7];
  q[0] = lo; q[1] = hi;
}

}  // namespace sbn32
