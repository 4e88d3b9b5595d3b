// sumcheck_comb_kernels.cuh — the "par" instances of prove_cubic_batched with the coeffs combination folded in.
//
// Reference: SumcheckInstanceProof::prove_cubic_batched, src/sumcheck.rs:165-330.  The transcript never sees the per-instance
// sums (e0, e2, e3)_i of the "par" instances (:201-235), only sum_i coeffs[i] * (e0, e2, e3)_i (:269-271).  With the shared
// third table C factored out,
//        sum_i c_i sum_x A_i(x,t) B_i(x,t) C(x,t)  =  sum_x C(x,t) * [ sum_i (c_i A_i)(x,t) * B_i(x,t) ],
// the bracket is a sum of products of two field elements: its 17 product columns are added up in 64-bit registers and the Montgomery
// reduction runs ONCE per index for all instances (fp.cuh: Cols).  c_i is folded into A_i when the table is first bound
// (A'_i = c_i A_i; binding is linear, so every later round sees c_i * bound(A_i) and the final A_i[0] is A'_i[0] / c_i, one host
// inversion per sumcheck — the same field element, bit for bit).  Per "par" instance and index pair the evaluation costs 3 x 81
// multiplier instructions instead of 6 x 171; the four binds (one product per stored element: the minimum) stay.
//
// The bracket at the three points comes from three running column sums over the instances,
//        LL = sum lo_a lo_b,   HH = sum hi_a hi_b,   DD = sum (hi_a - lo_a)(hi_b - lo_b)        (lo = Z'[i], hi = Z'[i + q])
//        S(0) = LL,  S(2) = 2 HH - LL + 2 DD,  S(3) = 3 HH - 2 LL + 6 DD                       (p(t) = lo + t (hi - lo), sumcheck.rs:111-135)
// so no evaluation point of A or B is ever formed.  Results are identical to the per-instance kernels' (field arithmetic is exact).
#pragma once
#include "sumcheck_kernels.cuh"

namespace sbn {

constexpr int SC_COMB_MAX = 16;          // "par" instances one thread walks (the ops proof has 12, the mem proof 4: sparse_mlpoly_full.rs:1380-1415)
// One group of instances sharing C, in device memory (read with uniform loads).  Every group of a sumcheck — the round-0 sums and one set
// per planned combined round — is written when the sumcheck begins (one upload); the first bind's u, v are finished on the device
// (k_sc_first_uv) once its challenge is known.
struct ScCombGroup {
  const uint32_t* a_src[SC_COMB_MAX];
  const uint32_t* b_src[SC_COMB_MAX];
  uint32_t* a_dst[SC_COMB_MAX];
  uint32_t* b_dst[SC_COMB_MAX];
  const uint32_t* c;                     // bind+eval kernel: the shared C ALREADY bound to this round's challenge (c[i], c[i + q]); eval kernel: C itself
  const uint32_t* c_src;                 // bind+eval kernel, when not null: the shared C BEFORE this round's bind (4 q entries) — the thread binds its two entries itself
  uint32_t* c_dst;                       //   ... and (one group per launch: the writer) stores them here; `c` is not read then
  uint32_t n;                            // instances in the group
  uint32_t scale;                        // 1: the A tables are still unscaled: the first bind writes A' = u z0 + v z2 (u = c_i (1 - r), v = c_i r); eval: A' = u A
  uint32_t u[SC_COMB_MAX][8], v[SC_COMB_MAX][8];   // Montgomery form, canonical
};

__device__ __forceinline__ Fr fr_load_uniform(const uint32_t* p) {       // 8 words at a wave-uniform address
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = p[k];
  return fe_unpack<FrP>(w);
}
// a z0 + b z2 with ONE reduction: the scaled bind c (z0 + r (z2 - z0)) = (c (1 - r)) z0 + (c r) z2.  u, v canonical; z below 2.5 r:
// the sum is below 5 r^2, the result below 1.03 r + r — stored as it is (a table value below 2.5 r)
__device__ __forceinline__ Fr sc_bind_scaled(const Fr& z0, const Fr& z2, const Fr& u, const Fr& v) {
  Cols s; cols_zero(s);
  cols_mac<FrP>(s, u, z0); cols_mac<FrP>(s, v, z2);
  return cols_reduce<FrP>(s);
}

// S(0), S(2), S(3) of the bracket from the reduced column sums L, H, D: normalised, and for 16 instances of table values below 2.5 r
// (differences + 3 r below 5.5 r) below 16 * 6.25 / 169 + 1 = 1.6 r (L, H) and 16 * 30.25 / 169 + 1 = 3.9 r (D).
// All on the unsigned path: U = H + D (< 5.5 r), S2 = 2 U - L + 2 r (< 13 r), V = U + D (< 9.4 r), S3 = 3 V - 2 L + 4 r (< 32.2 r);
// times C's points (below 2.5 r, 8 r, 13.5 r) the terms come back below 3.6 r: fr_acc32.
struct ScS { Fr s0, s2, s3; };
__device__ __forceinline__ ScS sc_bracket_points(const Fr& L, const Fr& H, const Fr& D) {
  ScS o; o.s0 = L;
  const Fr U = fe_normu(fe_add_lazy(H, D));
  o.s2 = fe_normu(fe_add_lazy(fe_dbl_lazy(U), fe_negb<FrP, 2>(L)));                      // 2 U + (2 r - L): limbs below 2^30 + 2^30.6
  const Fr V = fe_normu(fe_add_lazy(U, D));
  Fr V3;
#pragma unroll
  for (int k = 0; k < NL; k++) V3.v[k] = 3u * V.v[k];
  o.s3 = fe_normu(fe_add_lazy(V3, fe_subb<FrP, 4, 2>(fe_zero<FrP>(), fe_dbl_lazy(L))));  // 3 V + (4 r - 2 L): 2 L is "two normalised values", below 3 r
  return o;
}

// The three running sums of a thread live in LDS, not in registers (27 registers the column sums need: with them in registers the
// bind+eval kernel spilled at 2 waves per SIMD).  One read-add-write of 9 words per point and INDEX (not per instance).
struct ScLdsAcc { uint32_t v[3][NL][256]; };
__device__ __forceinline__ void sc_lds_acc_zero(ScLdsAcc& A) {
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int k = 0; k < NL; k++) A.v[t][k][threadIdx.x] = 0;
}
__device__ __forceinline__ Fr sc_lds_acc_get(const ScLdsAcc& A, int t) { Fr x;
#pragma unroll
  for (int k = 0; k < NL; k++) x.v[k] = A.v[t][k][threadIdx.x];
  return x; }
// acc[t] += x (terms below 4 r; a reduction every 32 terms keeps the value below fe_reduce's 169 r)
__device__ __forceinline__ void sc_lds_acc_add(ScLdsAcc& A, int t, const Fr& x, uint32_t cnt) {
  Fr a = fe_add(sc_lds_acc_get(A, t), x);
  if ((cnt & 31u) == 31u) a = fe_reduce(a);
#pragma unroll
  for (int k = 0; k < NL; k++) A.v[t][k][threadIdx.x] = a.v[k];
}

// ---- the streaming fused round of the "par" group: bind A', B to r_j and accumulate round j+1's COMBINED sums ------------------
// One thread walks all instances of an index: per instance 4 + 4 loads (quarter q of the old length apart), 4 binds, 4 stores, three
// 81-multiplier column accumulations.  The next table's four elements are in flight while the current one is bound (the same
// software pipeline as k_sc_bind_eval_pf).  Output: the block's three sums to partial[(slot * gridDim.x + blockIdx.x) * 3 + ..], folded
// by the last block into mailbox slot `slot` (sc_last_block_fold's protocol; slot = the "instance" index the host waits on).
template <bool SCALE>
__device__ __forceinline__ void sc_comb_bind_eval_body(const ScCombGroup* __restrict__ g, size_t q, const Fr& r, uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets,
                                                       uint32_t* __restrict__ out, uint32_t slot, uint32_t seq, uint32_t bidx, uint32_t nblk) {
  const uint32_t n = g->n;
  const uint32_t* csrc = g->c_src; uint32_t* cdst = g->c_dst;
  __shared__ ScLdsAcc acc;
  sc_lds_acc_zero(acc);
  uint32_t cnt = 0;
#define SC_PIN() __builtin_amdgcn_sched_barrier(0)
  for (size_t i = (size_t)bidx * blockDim.x + threadIdx.x; i < q; i += (size_t)nblk * blockDim.x) {
    Cols LL, HH, DD; cols_zero(LL); cols_zero(HH); cols_zero(DD);
    ScQuad zn = sc_quad_load<false>(g->a_src[0], i, q);
    uint32_t pend = 0;                                   // products in the column sums since the last carry pass
    for (uint32_t k = 0; k < n; k++) {
      ScPair a, b;
      {
        const ScQuad z = zn; zn = sc_quad_load<false>(g->b_src[k], i, q); SC_PIN();
        if (SCALE) {
          const Fr u = fr_load_uniform(g->u[k]), v = fr_load_uniform(g->v[k]);
          a.lo = sc_bind_scaled(z.z0, z.z2, u, v); a.hi = sc_bind_scaled(z.z1, z.z3, u, v);
        } else { a.lo = sc_bind1(z.z0, z.z2, r); a.hi = sc_bind1(z.z1, z.z3, r); }
        uint32_t* d = g->a_dst[k];
        fe_gstore_packed<FrP>(d + 8 * i, a.lo); fe_gstore_packed<FrP>(d + 8 * (i + q), a.hi);
      }
      {
        const ScQuad z = zn;
        const uint32_t kn = k + 1 < n ? k + 1 : k;       // (the last step re-loads its own A: 4 of ~100 loads, L2 hits)
        zn = sc_quad_load<false>(g->a_src[kn], i, q); SC_PIN();
        b.lo = sc_bind1(z.z0, z.z2, r); b.hi = sc_bind1(z.z1, z.z3, r);
        uint32_t* d = g->b_dst[k];
        fe_gstore_packed<FrP>(d + 8 * i, b.lo); fe_gstore_packed<FrP>(d + 8 * (i + q), b.hi);
      }
      if (pend == 6) { cols_carry(LL); cols_carry(HH); cols_carry(DD); pend = 0; }
      pend++;
      const Fr da = fe_normu(fe_subb<FrP, 3, 1>(a.hi, a.lo)), db = fe_normu(fe_subb<FrP, 3, 1>(b.hi, b.lo));
      cols_mac<FrP>(LL, a.lo, b.lo); cols_mac<FrP>(HH, a.hi, b.hi); cols_mac<FrP>(DD, da, db);
    }
    // the shared C of this index: either bound ahead of the launch (g->c) or bound HERE from the unbound table (g->c_src: two more
    // products per index instead of a launch of its own per round); its loads are issued ahead of the three reductions.  Not in the
    // scaling first bind (SCALE): that variant sits at the register limit (256 VGPRs) and would spill.
    const bool fuse = !SCALE && csrc != nullptr;
    ScQuad zc;
    if (fuse) zc = sc_quad_load<false>(csrc, i, q); else zc = sc_quad_load<true>(g->c, i, q);
    SC_PIN();
    const ScS S = sc_bracket_points(cols_reduce<FrP>(LL), cols_reduce<FrP>(HH), cols_reduce<FrP>(DD));
    Fr cl = zc.z0, ch = zc.z1;
    if (fuse) {
      cl = sc_bind1(zc.z0, zc.z2, r); ch = sc_bind1(zc.z1, zc.z3, r);
      if (cdst) { fe_gstore_packed<FrP>(cdst + 8 * i, cl); fe_gstore_packed<FrP>(cdst + 8 * (i + q), ch); }
    }
    const ScPts pc = sc_points_u(cl, ch);
    sc_lds_acc_add(acc, 0, fe_mulu(S.s0, cl), cnt); sc_lds_acc_add(acc, 1, fe_mulu(S.s2, pc.v2), cnt); sc_lds_acc_add(acc, 2, fe_mulu(S.s3, pc.v3), cnt);
    cnt++;
  }
#undef SC_PIN
  // the block's triple -> partial[slot][block], then the ticketed fold into the mailbox (sumcheck_kernels.cuh)
  sc_block_sums_store_at(sc_lds_acc_get(acc, 0), sc_lds_acc_get(acc, 1), sc_lds_acc_get(acc, 2), partial, slot, bidx, nblk);
  sc_last_block_fold_at(partial, tickets, out, 3, seq, slot, nblk);
}
template <bool SCALE>
__global__ void __launch_bounds__(256, 2) k_sc_comb_bind_eval(const ScCombGroup* __restrict__ groups, size_t q, ScScalar rmont, uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets,
                                                              uint32_t* __restrict__ out, uint32_t slot0, uint32_t seq) {
  sc_comb_bind_eval_body<SCALE>(groups + blockIdx.y, q, fr_from_words(rmont), partial, tickets, out, slot0 + blockIdx.y, seq, blockIdx.x, gridDim.x);
}

// ---- one launch for a whole streaming round: the "par" groups AND the "seq" instances ------------------------------------------------
// The combined kernel above is bound by its memory pattern (VALU ~40 % busy per wave), the per-instance streaming kernel of the "seq"
// instances by the VALU (12 products per index pair).  Launched one after the other they add up; in ONE grid, with the two kinds of
// block interleaved so that a CU holds one of each (both are 2-blocks-per-CU kernels), the VALU-bound blocks run in the memory-bound
// blocks' shadow.  Block b of the 1-D grid is a "seq" block when floor((b + 1) Ts / T) > floor(b Ts / T) (Ts of the T blocks, evenly
// spread); its index among its kind is dealt round-robin over the instances / groups.  Slots: seq instance j -> j, group g -> n_seq + g.
template <bool SCALE>
__global__ void __launch_bounds__(256, 2) k_sc_round_mixed(const ScCombGroup* __restrict__ groups, uint32_t n_groups, uint32_t gx_comb, ScFusedPack seqpack, uint32_t n_seq, uint32_t gx_seq, size_t q,
                                                           ScScalar rmont, uint32_t* __restrict__ partial_seq, uint32_t* __restrict__ partial_comb, uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, uint32_t seq,
                                                           unsigned long long* __restrict__ dbg /* null, or 4 words per block: kind, XCC id, start, end (100 MHz clock) — SBN_SC_DEBUG_BLOCKS */) {
  const unsigned long long Ts = (unsigned long long)n_seq * gx_seq, T = Ts + (unsigned long long)n_groups * gx_comb, b = blockIdx.x;
  const unsigned long long s0 = b * Ts / T, s1 = (b + 1) * Ts / T;
  const Fr r = fr_from_words(rmont);
  const unsigned long long t_begin = dbg ? wall_clock64() : 0ull;
  if (s1 > s0) {
    const uint32_t inst = (uint32_t)(s0 % n_seq), bidx = (uint32_t)(s0 / n_seq);
    ScFusedArgs a;
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)inst) a = seqpack.a[i];
    Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
    sc_pf_body<KIND_CUBIC, 0>(a, q, r, e0, e2, e3, bidx, gx_seq);
    sc_block_sums_store_slot(e0, e2, e3, true, partial_seq, nullptr, 0u, inst, bidx, gx_seq);
    sc_last_block_fold_at(partial_seq, tickets, out, 3, seq, inst, gx_seq);
  } else {
    const uint32_t ci = (uint32_t)(b - s0), grp = ci % n_groups, bidx = ci / n_groups;
    sc_comb_bind_eval_body<SCALE>(groups + grp, q, r, partial_comb, tickets, out, n_seq + grp, seq, bidx, gx_comb);
  }
  if (dbg && threadIdx.x == 0) {
    dbg[4 * b] = s1 > s0 ? 1ull : 0ull;
    dbg[4 * b + 1] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);      // HW_REG_XCC_ID[3:0]
    dbg[4 * b + 2] = t_begin; dbg[4 * b + 3] = wall_clock64();
  }
}


// ---- round 0 of the "par" group: the combined sums of the UNBOUND tables (no challenge yet) ---------------------------------------
// A is scaled in registers when the tables are still unscaled (scale != 0: two products per instance), C = g->c is the table itself.
__device__ __forceinline__ void sc_comb_eval_body(const ScCombGroup* __restrict__ g, size_t half, Fr& e0, Fr& e2, Fr& e3, uint32_t bidx, uint32_t nblk) {
  const uint32_t n = g->n;
  const bool scale = g->scale != 0;
  uint32_t c0 = 0, c2 = 0, c3 = 0;
  for (size_t i = (size_t)bidx * blockDim.x + threadIdx.x; i < half; i += (size_t)nblk * blockDim.x) {
    Cols LL, HH, DD; cols_zero(LL); cols_zero(HH); cols_zero(DD);
    ScQuad zn;
    zn.z0 = fe_gload<FrP>(g->a_src[0] + 8 * i); zn.z1 = fe_gload<FrP>(g->a_src[0] + 8 * (i + half));
    zn.z2 = fe_gload<FrP>(g->b_src[0] + 8 * i); zn.z3 = fe_gload<FrP>(g->b_src[0] + 8 * (i + half));
    uint32_t pend = 0;
    for (uint32_t k = 0; k < n; k++) {
      const ScQuad z = zn;
      const uint32_t kn = k + 1 < n ? k + 1 : k;
      zn.z0 = fe_gload<FrP>(g->a_src[kn] + 8 * i); zn.z1 = fe_gload<FrP>(g->a_src[kn] + 8 * (i + half));
      zn.z2 = fe_gload<FrP>(g->b_src[kn] + 8 * i); zn.z3 = fe_gload<FrP>(g->b_src[kn] + 8 * (i + half));
      __builtin_amdgcn_sched_barrier(0);
      Fr alo = z.z0, ahi = z.z1;
      if (scale) { const Fr u = fr_load_uniform(g->u[k]); alo = fe_mulu(u, alo); ahi = fe_mulu(u, ahi); }
      if (pend == 6) { cols_carry(LL); cols_carry(HH); cols_carry(DD); pend = 0; }
      pend++;
      const Fr da = fe_normu(fe_subb<FrP, 3, 1>(ahi, alo)), db = fe_normu(fe_subb<FrP, 3, 1>(z.z3, z.z2));
      cols_mac<FrP>(LL, alo, z.z2); cols_mac<FrP>(HH, ahi, z.z3); cols_mac<FrP>(DD, da, db);
    }
    const Fr cl = fe_gload<FrP>(g->c + 8 * i), ch = fe_gload<FrP>(g->c + 8 * (i + half));
    const ScS S = sc_bracket_points(cols_reduce<FrP>(LL), cols_reduce<FrP>(HH), cols_reduce<FrP>(DD));
    const ScPts pc = sc_points_u(cl, ch);
    fr_acc32(e0, fe_mulu(S.s0, cl), c0); fr_acc32(e2, fe_mulu(S.s2, pc.v2), c2); fr_acc32(e3, fe_mulu(S.s3, pc.v3), c3);
  }
}
__global__ void __launch_bounds__(256, 2) k_sc_comb_eval(const ScCombGroup* __restrict__ groups, size_t half, uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets,
                                                         uint32_t* __restrict__ out, uint32_t slot0, uint32_t seq) {
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  sc_comb_eval_body(groups + blockIdx.y, half, e0, e2, e3, blockIdx.x, gridDim.x);
  sc_block_sums_store_at(e0, e2, e3, partial, slot0 + blockIdx.y, blockIdx.x, gridDim.x);
  sc_last_block_fold_at(partial, tickets, out, 3, seq, slot0 + blockIdx.y, gridDim.x);
}

// Round 0 of a sumcheck with both kinds of instance as ONE launch (as k_sc_round_mixed does for the binds): the "seq" instances'
// cubic sums (the loop of k_sc_eval<KIND_CUBIC>, 6 products per index: VALU-heavy) interleaved with the "par" groups' combined sums.
// Slots: seq instance j -> j, group g -> n_seq + g.
__global__ void __launch_bounds__(256, 2) k_sc_eval_mixed(const ScCombGroup* __restrict__ groups, uint32_t n_groups, uint32_t gx_comb, ScArgsPack seqpack, uint32_t n_seq, uint32_t gx_seq, size_t half,
                                                          uint32_t* __restrict__ partial_seq, uint32_t* __restrict__ partial_comb, uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, uint32_t seq) {
  const unsigned long long Ts = (unsigned long long)n_seq * gx_seq, T = Ts + (unsigned long long)n_groups * gx_comb, b = blockIdx.x;
  const unsigned long long s0 = b * Ts / T, s1 = (b + 1) * Ts / T;
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  if (s1 > s0) {
    const uint32_t inst = (uint32_t)(s0 % n_seq), bidx = (uint32_t)(s0 / n_seq);
    ScArgs a;
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)inst) a = seqpack.a[i];
    uint32_t c0 = 0, c2 = 0, c3 = 0;
    for (size_t i = (size_t)bidx * blockDim.x + threadIdx.x; i < half; i += (size_t)gx_seq * blockDim.x) {
      const Fr al = fe_gload<FrP>(a.t[0] + 8 * i), ah = fe_gload<FrP>(a.t[0] + 8 * (i + half));
      const Fr bl = fe_gload<FrP>(a.t[1] + 8 * i), bh = fe_gload<FrP>(a.t[1] + 8 * (i + half));
      const Fr cl = fe_gload<FrP>(a.t[2] + 8 * i), ch = fe_gload<FrP>(a.t[2] + 8 * (i + half));
      const ScPts pa = sc_points_u(al, ah), pb = sc_points_u(bl, bh), pc = sc_points_u(cl, ch);
      fr_acc(e0, fe_mulu(fe_mulu(al, bl), cl), c0);
      fr_acc(e2, fe_mulu(fe_mulu(pa.v2, pb.v2), pc.v2), c2);
      fr_acc(e3, fe_mulu(fe_mulu(pa.v3, pb.v3), pc.v3), c3);
    }
    sc_block_sums_store_at(e0, e2, e3, partial_seq, inst, bidx, gx_seq);
    sc_last_block_fold_at(partial_seq, tickets, out, 3, seq, inst, gx_seq);
  } else {
    const uint32_t ci = (uint32_t)(b - s0), grp = ci % n_groups, bidx = ci / n_groups;
    sc_comb_eval_body(groups + grp, half, e0, e2, e3, bidx, gx_comb);
    sc_block_sums_store_at(e0, e2, e3, partial_comb, n_seq + grp, bidx, gx_comb);
    sc_last_block_fold_at(partial_comb, tickets, out, 3, seq, n_seq + grp, gx_comb);
  }
}

}  // namespace sbn
