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
// One group of instances sharing C, in device memory (read with uniform loads).  Three of these are written when a sumcheck
// begins (ping, pong, and the first bind, which also scales) — nothing is uploaded per round except the first bind's u, v.
struct ScCombGroup {
  const uint32_t* a_src[SC_COMB_MAX];
  const uint32_t* b_src[SC_COMB_MAX];
  uint32_t* a_dst[SC_COMB_MAX];
  uint32_t* b_dst[SC_COMB_MAX];
  const uint32_t* c;                     // bind+eval kernel: the shared C ALREADY bound to this round's challenge (c[i], c[i + q]); eval kernel: C itself
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
__global__ void __launch_bounds__(256, 2) k_sc_comb_bind_eval(const ScCombGroup* __restrict__ groups, size_t q, ScScalar rmont, uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets,
                                                              uint32_t* __restrict__ out, uint32_t slot0, uint32_t seq) {
  const ScCombGroup* __restrict__ g = groups + blockIdx.y;
  const Fr r = fr_from_words(rmont);
  const uint32_t n = g->n;
  __shared__ ScLdsAcc acc;
  sc_lds_acc_zero(acc);
  uint32_t cnt = 0;
#define SC_PIN() __builtin_amdgcn_sched_barrier(0)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
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
    const Fr cl = fe_gload<FrP>(g->c + 8 * i), ch = fe_gload<FrP>(g->c + 8 * (i + q));
    const ScS S = sc_bracket_points(cols_reduce<FrP>(LL), cols_reduce<FrP>(HH), cols_reduce<FrP>(DD));
    const ScPts pc = sc_points_u(cl, ch);
    sc_lds_acc_add(acc, 0, fe_mulu(S.s0, cl), cnt); sc_lds_acc_add(acc, 1, fe_mulu(S.s2, pc.v2), cnt); sc_lds_acc_add(acc, 2, fe_mulu(S.s3, pc.v3), cnt);
    cnt++;
  }
#undef SC_PIN
  // the block's triple -> partial[slot][block], then the ticketed fold into the mailbox (sumcheck_kernels.cuh)
  sc_block_sums_store_at(sc_lds_acc_get(acc, 0), sc_lds_acc_get(acc, 1), sc_lds_acc_get(acc, 2), partial, slot0 + blockIdx.y);
  sc_last_block_fold_at(partial, tickets, out, 3, seq, slot0 + blockIdx.y);
}

// ---- round 0 of the "par" group: the combined sums of the UNBOUND tables (no challenge yet) ---------------------------------------
// A is scaled in registers when the tables are still unscaled (scale != 0: two products per instance), C = g->c is the table itself.
__global__ void __launch_bounds__(256, 2) k_sc_comb_eval(const ScCombGroup* __restrict__ groups, size_t half, uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets,
                                                         uint32_t* __restrict__ out, uint32_t slot0, uint32_t seq) {
  const ScCombGroup* __restrict__ g = groups + blockIdx.y;
  const uint32_t n = g->n;
  const bool scale = g->scale != 0;
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  uint32_t c0 = 0, c2 = 0, c3 = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
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
  sc_block_sums_store_at(e0, e2, e3, partial, slot0 + blockIdx.y);
  sc_last_block_fold_at(partial, tickets, out, 3, seq, slot0 + blockIdx.y);
}


// ---- the tail of a sumcheck: ALL remaining rounds in ONE launch ------------------------------------------------------------------
// From tables of SC_TAIL_MAX_LEN entries down a round is a host round trip around a few microseconds of arithmetic: as separate
// launches each costs ~10 us of kernel (launch, fold, flag) + ~16 us of gap (measured: 352 such rounds per keyless-shaped prove).
// Here one block per instance copies its three tables into LDS once and then loops: bind to r_j in LDS, sum the next round, publish
// the triple to the host mailbox, poll the mailbox for r_{j+1}.  No launch, no global-memory round trip, no cross-block hand-over per
// round (the "par" instances each keep their own copy of the shared C: binding it twelve times costs nothing at this size).
// Host <-> kernel protocol (coherent pinned memory, tail mailbox of the context):
//   results   tm[slot * 24 ..]  three canonical scalars per instance, then  tm[TAIL_FLAGS + slot] = seq   (sc_flag_store: drained + released)
//   challenge tm[TAIL_CHAL + 2 ..] = r in the tables' Montgomery form, then tm[TAIL_CHAL] = seq            (host: release store)
//   abort     tm[TAIL_CHAL + 1] != 0: every block leaves at its next poll (the state was freed, or an error on the host side)
// Every wait is bounded: a block that sees no challenge for SC_TAIL_TIMEOUT_TICKS of the 100 MHz real-time counter raises
// tm[TAIL_CHAL + 10] and leaves — the grid always drains, whatever the host does.
constexpr int SC_TAIL_MAX_LEN = 1024;                  // 3 tables x 1024 x 32 B = 96 KiB of LDS
constexpr int TAIL_FLAGS = SC_PACK_MAX * 24;           // word offsets inside the tail mailbox
constexpr int TAIL_FINALS = TAIL_FLAGS + SC_PACK_MAX + 8;      // per instance A[0], B[0], C[0] (canonical), then TAIL_FLAGS + SC_PACK_MAX = their flag
constexpr int TAIL_CHAL = TAIL_FINALS + SC_PACK_MAX * 24;      // seq, abort, r[8], timed_out
constexpr int TAIL_WORDS = TAIL_CHAL + 16;
constexpr unsigned long long SC_TAIL_TIMEOUT_TICKS = 1000000000ull;   // 10 s at 100 MHz
struct ScTailPack { const uint32_t* a[SC_PACK_MAX]; const uint32_t* b[SC_PACK_MAX]; const uint32_t* c[SC_PACK_MAX]; ScScalar scale[SC_PACK_MAX]; uint32_t has_scale[SC_PACK_MAX]; };

__device__ __forceinline__ Fr fr_lds_get(const uint32_t* s, size_t i) {       // 8 words of element i (LDS, two 16-byte reads)
  const uint4* q = reinterpret_cast<const uint4*>(s + 8 * i);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return fe_unpack<FrP>(w);
}
__device__ __forceinline__ void fr_lds_put(uint32_t* s, size_t i, const Fr& a) {
  uint32_t w[8]; fe_pack<FrP>(a, w);
  uint4* q = reinterpret_cast<uint4*>(s + 8 * i);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// eval_first != 0: the tables are fresh (round 0): publish their sums before the first challenge; otherwise r0 is the challenge of the
// round that hands over (its sums were delivered by the launch before).  seq0: the sequence number of the first publication.
__global__ void __launch_bounds__(512) k_sc_tail(ScTailPack pack, uint32_t len, ScScalar r0_mont, uint32_t eval_first, uint32_t* __restrict__ tm, uint32_t seq0) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];      // [3][len][8]
  __shared__ uint32_t sm[8][3][NL];
  __shared__ uint32_t s_ctl[12];                        // challenge words + go / abort
  const int inst = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t* src[3] = {nullptr, nullptr, nullptr}; ScScalar sc; uint32_t has_scale = 0;
#pragma unroll
  for (int i = 0; i < SC_PACK_MAX; i++) if (i == inst) { src[0] = pack.a[i]; src[1] = pack.b[i]; src[2] = pack.c[i]; sc = pack.scale[i]; has_scale = pack.has_scale[i]; }
  uint32_t* T[3] = {s_tab, s_tab + 8 * (size_t)len, s_tab + 16 * (size_t)len};
  {
    const Fr scale = fr_from_words(sc);
    for (int t = 0; t < 3; t++)
      for (size_t i = tid; i < len; i += blockDim.x) {
        Fr x = fe_gload<FrP>(src[t] + 8 * i);
        if (t == 0 && has_scale) x = fe_mulu(scale, x);                       // c_i folded into A as the table comes in
        fr_lds_put(T[t], i, x);
      }
  }
  __syncthreads();
  Fr r = fr_from_words(r0_mont);
  uint32_t seq = seq0;
  bool need_bind = eval_first == 0;
  for (;;) {
    if (need_bind) {
      const size_t half = len / 2;
      for (size_t i = tid; i < half; i += blockDim.x)
#pragma unroll 1
        for (int t = 0; t < 3; t++) fr_lds_put(T[t], i, sc_bind1(fr_lds_get(T[t], i), fr_lds_get(T[t], i + half), r));     // in place: index i < half only
      len = (uint32_t)half;
      __syncthreads();
    }
    need_bind = true;
    if (len == 1) break;
    // the sums of this round (sumcheck.rs:201-235 for one instance)
    const size_t h = len / 2;
    Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
    for (size_t i = tid; i < h; i += blockDim.x) {
      const Fr al = fr_lds_get(T[0], i), ah = fr_lds_get(T[0], i + h), bl = fr_lds_get(T[1], i), bh = fr_lds_get(T[1], i + h), cl = fr_lds_get(T[2], i), ch = fr_lds_get(T[2], i + h);
      const ScPts pa = sc_points_u(al, ah), pb = sc_points_u(bl, bh), pc = sc_points_u(cl, ch);
      e0 = fe_add(e0, fe_mulu(fe_mulu(al, bl), cl)); e2 = fe_add(e2, fe_mulu(fe_mulu(pa.v2, pb.v2), pc.v2)); e3 = fe_add(e3, fe_mulu(fe_mulu(pa.v3, pb.v3), pc.v3));
    }
    e0 = wave_sum_fr(fe_reduce(e0)); e2 = wave_sum_fr(fe_reduce(e2)); e3 = wave_sum_fr(fe_reduce(e3));
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NL; k++) { sm[wv][0][k] = e0.v[k]; sm[wv][1][k] = e2.v[k]; sm[wv][2][k] = e3.v[k]; }
    }
    __syncthreads();
    if (tid < 3) {
      Fr s = fe_zero<FrP>();
      for (int w = 0; w < (int)(blockDim.x >> 6); w++) { Fr x; for (int k = 0; k < NL; k++) x.v[k] = sm[w][tid][k]; s = fe_add(s, x); }
      fe_store_packed<FrP>(tm + 8 * ((size_t)inst * 3 + tid), fe_from_mont(s));
    }
    sc_drain_stores();
    __syncthreads();
    if (tid == 0) {
      sc_flag_store(tm + TAIL_FLAGS + inst, seq);
      // wait for the next challenge (bounded)
      seq++;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      uint32_t go = 0;
      for (;;) {
        if (__hip_atomic_load(tm + TAIL_CHAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == seq) { go = 1; break; }
        if (__hip_atomic_load(tm + TAIL_CHAL + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SC_TAIL_TIMEOUT_TICKS) { __hip_atomic_store(tm + TAIL_CHAL + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(8);
      }
      if (go) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        for (int k = 0; k < 8; k++) s_ctl[k] = __hip_atomic_load(tm + TAIL_CHAL + 2 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      s_ctl[8] = go;
    }
    __syncthreads();
    if (!s_ctl[8]) return;                                 // abort or time-out: every thread of the block leaves here
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = s_ctl[k];
    r = fe_unpack<FrP>(w);
    __syncthreads();                                       // s_ctl is rewritten in the next round
  }
  // final claims: A[0], B[0], C[0] of this instance (sumcheck.rs:302-318)
  if (tid < 3) fe_store_packed<FrP>(tm + TAIL_FINALS + 8 * ((size_t)inst * 3 + tid), fe_from_mont(fr_lds_get(T[tid], 0)));
  sc_drain_stores();
  __syncthreads();
  if (tid == 0) sc_flag_store(tm + TAIL_FLAGS + inst, seq);
}

}  // namespace sbn
