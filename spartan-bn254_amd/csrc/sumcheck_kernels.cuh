// sumcheck_kernels.cuh — per-round multilinear evaluations and the top-variable bind, over BN254 Fr.
//
// Reference loops replaced (all serial `for i in 0..len` loops over Vec<Scalar> there):
//   prove_cubic                     src/sumcheck.rs:111-135   f = A*B*C            -> k_sc_eval<KIND_CUBIC>
//   prove_cubic_batched             src/sumcheck.rs:201-267   same, many instances  -> grid.y = instance
//   prove_cubic_with_additive_term  src/sumcheck.rs:502-530   f = T*(A*B - C)       -> k_sc_eval<KIND_R1CS>
//   prove_quad                      src/sumcheck.rs:691-699   f = Z*ABC (pts 0,2)   -> k_sc_eval<KIND_QUAD>
//   bound_poly_var_top              src/hyrax.rs:195-203      Z[i] += r*(Z[i+n]-Z[i]) -> k_bind_top
//   EqPolynomial::evals             src/hyrax.rs:355-369                              -> k_eq_level
// Evaluation points follow the reference exactly: p(2) = 2*hi - lo, p(3) = p(2) + hi - lo, and the host
// receives (e0, e2, e3) — e1 is derived from the claim on the host (sumcheck.rs:137).
//
// Tables are AoS arrays of 32-byte Montgomery-form elements; lane i reads element i with two 16-byte loads, so
// a wavefront reads 2 KiB contiguous per table half — the kernels are HBM-streaming with ~1 product per 32 B.
// Field addition is associative and commutative mod r, so the tree-shaped partial sums give the same canonical
// result as the reference's left-to-right loop.
#pragma once
#include "fp.cuh"

namespace sbn {

enum { KIND_CUBIC = 0, KIND_R1CS = 1, KIND_QUAD = 2 };

struct ScArgs {          // one instance: up to 4 tables, all of length 2*half
  const uint32_t* t[4];
};

__device__ __forceinline__ Fr wave_sum_fr(Fr v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = __shfl_down(v.v[i], d, 64);
    v = fe_add(v, o);
  }
  return v;
}


// ---- single-launch rounds -------------------------------------------------------------------------------------------
// A sumcheck round is a host round trip (the next challenge comes from the transcript), and below ~2^16 entries the trip,
// not the data, is the cost.  So a round is ONE launch: the table pointers and the challenge travel as kernel arguments (no
// staging copy), and the last block of every instance to finish — found with a ticket counter — folds the per-block partial
// sums and stores the canonical values straight into host-visible pinned memory (no finishing kernel, no copy back).
constexpr int SC_PACK_MAX = 24;            // instances whose pointers fit in the kernel-argument block
struct ScScalar { uint32_t v[8]; };        // a canonical challenge passed by value

__device__ __forceinline__ Fr fr_load_coherent(const uint32_t* p) {   // written by other blocks of this launch: bypass the CU's L1
  Fr x;
#pragma unroll
  for (int k = 0; k < 8; k++) x.v[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return x;
}
// Called by all 256 threads after the block's triple went to partial[(inst * gridDim.x + blockIdx.x) * 3 + q].  nq = 2 or 3.
// `out` = mailbox in coherent pinned host memory: 24 x 96 B of results, then one flag word per instance; the flag is stored
// (release, system scope) after the results, so the host can poll it instead of paying a stream synchronisation.
constexpr int SC_MBOX_FLAGS = SC_PACK_MAX * 24;      // word offset of the flags inside the mailbox
__device__ __forceinline__ void sc_last_block_fold(const uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, int nq, uint32_t seq) {
  __shared__ uint32_t s_last;
  __threadfence();                         // the triple is visible device-wide before the ticket is taken
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&tickets[blockIdx.y], 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wv < 3) {
    Fr s = fe_zero<FrP>();
    if (wv < nq) for (unsigned b = lane; b < gridDim.x; b += 64) s = fe_add(s, fr_load_coherent(partial + 8 * (((size_t)blockIdx.y * gridDim.x + b) * 3 + wv)));
    s = wave_sum_fr(s);
    if (lane == 0) { fe_store<FrP>(out + 8 * ((size_t)blockIdx.y * 3 + wv), fe_from_mont(s)); __threadfence_system(); }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    tickets[blockIdx.y] = 0;                          // ready for the next launch
    __hip_atomic_store(out + SC_MBOX_FLAGS + blockIdx.y, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// partial[inst][block][3] (Montgomery form)
struct ScArgsPack { ScArgs a[SC_PACK_MAX]; };
// args == nullptr: the instances' pointers are in `pack`; tickets == nullptr: the caller runs k_sc_finish instead of the in-kernel fold
template <int KIND>
__global__ void __launch_bounds__(256) k_sc_eval(const ScArgs* __restrict__ args, ScArgsPack pack, size_t half, uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, uint32_t seq) {
  ScArgs a;
  if (args) a = args[blockIdx.y];
  else {
    // statically indexed selects: a dynamic index into a by-value argument would make the compiler copy the whole block to scratch
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  }
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    if (KIND == KIND_QUAD) {
      Fr zl = fe_gload<FrP>(a.t[0] + 8 * i), zh = fe_gload<FrP>(a.t[0] + 8 * (i + half));
      Fr al = fe_gload<FrP>(a.t[1] + 8 * i), ah = fe_gload<FrP>(a.t[1] + 8 * (i + half));
      e0 = fe_add(e0, fe_mul(zl, al));
      Fr z2 = fe_sub(fe_dbl(zh), zl), a2 = fe_sub(fe_dbl(ah), al);
      e2 = fe_add(e2, fe_mul(z2, a2));
    } else if (KIND == KIND_CUBIC) {
      Fr al = fe_gload<FrP>(a.t[0] + 8 * i), ah = fe_gload<FrP>(a.t[0] + 8 * (i + half));
      Fr bl = fe_gload<FrP>(a.t[1] + 8 * i), bh = fe_gload<FrP>(a.t[1] + 8 * (i + half));
      Fr cl = fe_gload<FrP>(a.t[2] + 8 * i), ch = fe_gload<FrP>(a.t[2] + 8 * (i + half));
      e0 = fe_add(e0, fe_mul(fe_mul(al, bl), cl));
      Fr da = fe_sub(ah, al), db = fe_sub(bh, bl), dc = fe_sub(ch, cl);
      Fr a2 = fe_add(ah, da), b2 = fe_add(bh, db), c2 = fe_add(ch, dc);        // 2*hi - lo
      e2 = fe_add(e2, fe_mul(fe_mul(a2, b2), c2));
      Fr a3 = fe_add(a2, da), b3 = fe_add(b2, db), c3 = fe_add(c2, dc);        // p(2) + hi - lo
      e3 = fe_add(e3, fe_mul(fe_mul(a3, b3), c3));
    } else {
      Fr tl = fe_gload<FrP>(a.t[0] + 8 * i), th = fe_gload<FrP>(a.t[0] + 8 * (i + half));
      Fr al = fe_gload<FrP>(a.t[1] + 8 * i), ah = fe_gload<FrP>(a.t[1] + 8 * (i + half));
      Fr bl = fe_gload<FrP>(a.t[2] + 8 * i), bh = fe_gload<FrP>(a.t[2] + 8 * (i + half));
      Fr cl = fe_gload<FrP>(a.t[3] + 8 * i), ch = fe_gload<FrP>(a.t[3] + 8 * (i + half));
      e0 = fe_add(e0, fe_mul(tl, fe_sub(fe_mul(al, bl), cl)));
      Fr dt = fe_sub(th, tl), da = fe_sub(ah, al), db = fe_sub(bh, bl), dc = fe_sub(ch, cl);
      Fr t2 = fe_add(th, dt), a2 = fe_add(ah, da), b2 = fe_add(bh, db), c2 = fe_add(ch, dc);
      e2 = fe_add(e2, fe_mul(t2, fe_sub(fe_mul(a2, b2), c2)));
      Fr t3 = fe_add(t2, dt), a3 = fe_add(a2, da), b3 = fe_add(b2, db), c3 = fe_add(c2, dc);
      e3 = fe_add(e3, fe_mul(t3, fe_sub(fe_mul(a3, b3), c3)));
    }
  }
  // block reduction: wave shuffles, then LDS across the 4 waves
  __shared__ uint32_t sm[4][3][8];
  e0 = wave_sum_fr(e0); e2 = wave_sum_fr(e2);
  if (KIND != KIND_QUAD) e3 = wave_sum_fr(e3);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) { sm[wv][0][k] = e0.v[k]; sm[wv][1][k] = e2.v[k]; sm[wv][2][k] = e3.v[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < 4; w++) { Fr x; for (int k = 0; k < 8; k++) x.v[k] = sm[w][threadIdx.x][k]; s = fe_add(s, x); }
    uint32_t* o = partial + 8 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x);
    for (int k = 0; k < 8; k++) o[k] = s.v[k];
  }
  if (tickets) sc_last_block_fold(partial, tickets, out, KIND == KIND_QUAD ? 2 : 3, seq);
}

// Fused round: bind every table to r_j (hyrax.rs:195-203) AND accumulate the round-(j+1) sums from the freshly bound
// values, in one pass: per table 4 elements are read (quarter q of the old length apart), the two bound values are
// written to the table's second buffer and feed the next round's evaluation points directly.  One pass per round
// instead of two: the separate eval pass (a full re-read of the bound tables) disappears.
//   dst[j] == nullptr : another instance writes that (shared) table; values are still computed for the sums.
//   pre[j] != 0       : src[j] is ALREADY bound (the "par" instances of prove_cubic_batched share one C table,
//                       sumcheck.rs:201-235: it is bound once by k_bind_oop ahead of this launch and every instance reads the two
//                       bound values instead of re-reading four and re-binding them: -2 of 12 products and -2 of 12 loads each).
// The tables of an instance are taken one after another and only the running products at the points 0, 2, 3 stay live, so the
// kernel fits 4 waves per SIMD; the challenge arrives in Montgomery form (one host product instead of one per thread), and the
// grid is sized for a few block rounds so that the wave/LDS reduction at the end is amortised over many indices per thread.
struct ScFusedArgs {
  const uint32_t* src[4];
  uint32_t* dst[4];
  uint32_t pre[4];
};
struct ScFusedPack { ScFusedArgs a[SC_PACK_MAX]; };


// one table at index i: bound values lo = Z'[i], hi = Z'[i + q]
struct ScPair { Fr lo, hi; };
__device__ __forceinline__ ScPair sc_bound_pair(const uint32_t* __restrict__ z, uint32_t* __restrict__ dst, uint32_t pre, size_t i, size_t q, const Fr& r) {
  ScPair o;
  if (pre) { o.lo = fe_gload<FrP>(z + 8 * i); o.hi = fe_gload<FrP>(z + 8 * (i + q)); return o; }
  const Fr z0 = fe_gload<FrP>(z + 8 * i), z2 = fe_gload<FrP>(z + 8 * (i + 2 * q));
  const Fr z1 = fe_gload<FrP>(z + 8 * (i + q)), z3 = fe_gload<FrP>(z + 8 * (i + 3 * q));
  o.lo = fe_add(z0, fe_mul(r, fe_sub(z2, z0)));
  o.hi = fe_add(z1, fe_mul(r, fe_sub(z3, z1)));
  if (dst) { fe_gstore<FrP>(dst + 8 * i, o.lo); fe_gstore<FrP>(dst + 8 * (i + q), o.hi); }
  return o;
}

// WPS = waves per SIMD the register allocation is held to (4 -> 128 VGPRs with a few spilled dwords; 3 -> no spills)
template <int KIND, int WPS>
__global__ void __launch_bounds__(256, WPS) k_sc_bind_eval(const ScFusedArgs* __restrict__ args, ScFusedPack pack, size_t q /* old_len / 4 */, ScScalar rmont /* Montgomery form */, uint32_t* __restrict__ partial,
                                                      uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, uint32_t seq) {
  ScFusedArgs a;
  if (args) a = args[blockIdx.y];
  else {
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  }
  Fr r; for (int k = 0; k < 8; k++) r.v[k] = rmont.v[k];
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
#define SC_TAB(j) sc_bound_pair(a.src[j], a.dst[j], a.pre[j], i, q, r)
    if (KIND == KIND_QUAD) {
      const ScPair z = SC_TAB(0);
      const Fr z2 = fe_sub(fe_dbl(z.hi), z.lo);
      const ScPair w = SC_TAB(1);
      e0 = fe_add(e0, fe_mul(z.lo, w.lo));
      e2 = fe_add(e2, fe_mul(z2, fe_sub(fe_dbl(w.hi), w.lo)));
    } else if (KIND == KIND_CUBIC) {
      Fr p0, p2, p3;
      { const ScPair t = SC_TAB(0); const Fr d = fe_sub(t.hi, t.lo); p0 = t.lo; p2 = fe_add(t.hi, d); p3 = fe_add(p2, d); }   // p(2) = 2 hi - lo, p(3) = p(2) + hi - lo
      { const ScPair t = SC_TAB(1); const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d); p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, v); p3 = fe_mul(p3, fe_add(v, d)); }
      { const ScPair t = SC_TAB(2); const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d); p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, v); p3 = fe_mul(p3, fe_add(v, d)); }
      e0 = fe_add(e0, p0); e2 = fe_add(e2, p2); e3 = fe_add(e3, p3);
    } else {
      // tau * (Az * Bz - Cz): tables 1, 2 (product), 3 (subtracted), then 0 (tau)
      Fr p0, p2, p3;
      { const ScPair t = SC_TAB(1); const Fr d = fe_sub(t.hi, t.lo); p0 = t.lo; p2 = fe_add(t.hi, d); p3 = fe_add(p2, d); }
      { const ScPair t = SC_TAB(2); const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d); p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, v); p3 = fe_mul(p3, fe_add(v, d)); }
      { const ScPair t = SC_TAB(3); const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d); p0 = fe_sub(p0, t.lo); p2 = fe_sub(p2, v); p3 = fe_sub(p3, fe_add(v, d)); }
      { const ScPair t = SC_TAB(0); const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d); p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, v); p3 = fe_mul(p3, fe_add(v, d)); }
      e0 = fe_add(e0, p0); e2 = fe_add(e2, p2); e3 = fe_add(e3, p3);
    }
#undef SC_TAB
  }
  __shared__ uint32_t sm[4][3][8];
  e0 = wave_sum_fr(e0); e2 = wave_sum_fr(e2);
  if (KIND != KIND_QUAD) e3 = wave_sum_fr(e3);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) { sm[wv][0][k] = e0.v[k]; sm[wv][1][k] = e2.v[k]; sm[wv][2][k] = e3.v[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < 4; w++) { Fr x; for (int k = 0; k < 8; k++) x.v[k] = sm[w][threadIdx.x][k]; s = fe_add(s, x); }
    uint32_t* o = partial + 8 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x);
    for (int k = 0; k < 8; k++) o[k] = s.v[k];
  }
  if (tickets) sc_last_block_fold(partial, tickets, out, KIND == KIND_QUAD ? 2 : 3, seq);
}

// The streaming form of the fused round (tables of 2^16 entries and more).  Measured on MI355X: the plain form above keeps too
// few bytes in flight — a wave issues the four loads of a table, waits, computes ~5 products, and only then issues the next
// table's loads, so with 3-4 waves per SIMD about 20 KB per CU are in flight on average and the round runs at ~2.9 TB/s with
// the multiplier 60 % busy (neither HBM nor the VALU saturated).  Here the NEXT table's elements are loaded into a second
// register set before the current table's arithmetic starts (software pipeline over the flattened (index, table) sequence):
// every wave has 8 KB in flight all the time.  The price is registers (2 waves per SIMD), which a VALU-dense loop without
// exposed load latency tolerates (tools/micro/ecbench.hip: 94 % of the mixed-addition ceiling at 2 waves per SIMD).
// PRE (which tables arrive pre-bound) is a compile-time mask: with run-time branches around the loads the compiler sinks every
// "prefetch" back down to its use (seen in the ISA), which is exactly the serial load -> wait -> compute chain this form removes.
// A table that is not pre-bound always has its writer here (the host pre-binds every shared table on this path).
struct ScQuad { Fr z0, z1, z2, z3; };     // the four elements of one table at index i (pre-bound table: z0 = lo, z1 = hi)
template <bool PRE> __device__ __forceinline__ ScQuad sc_quad_load(const uint32_t* __restrict__ z, size_t i, size_t q) {
  ScQuad o;
  o.z0 = fe_gload<FrP>(z + 8 * i); o.z1 = fe_gload<FrP>(z + 8 * (i + q));
  if (!PRE) { o.z2 = fe_gload<FrP>(z + 8 * (i + 2 * q)); o.z3 = fe_gload<FrP>(z + 8 * (i + 3 * q)); }
  return o;
}
template <bool PRE> __device__ __forceinline__ ScPair sc_quad_bind(const ScQuad& z, uint32_t* __restrict__ dst, size_t i, size_t q, const Fr& r) {
  ScPair o;
  if (PRE) { o.lo = z.z0; o.hi = z.z1; return o; }
  o.lo = fe_add(z.z0, fe_mul(r, fe_sub(z.z2, z.z0)));
  o.hi = fe_add(z.z1, fe_mul(r, fe_sub(z.z3, z.z1)));
  fe_gstore<FrP>(dst + 8 * i, o.lo); fe_gstore<FrP>(dst + 8 * (i + q), o.hi);
  return o;
}
// table order: cubic 0,1,2 — r1cs 1,2,3,0 (tau last: tau * (Az*Bz - Cz)) — quad 0,1.  PREMASK bit t = the t-th table IN THAT ORDER is pre-bound.
template <int KIND, int PREMASK>
__device__ __forceinline__ void sc_pf_body(const ScFusedArgs& a, size_t q, const Fr& r, Fr& e0, Fr& e2, Fr& e3) {
  constexpr int NT = KIND == KIND_QUAD ? 2 : KIND == KIND_CUBIC ? 3 : 4;
  constexpr int T0 = KIND == KIND_R1CS ? 1 : 0, T1 = KIND == KIND_R1CS ? 2 : 1, T2 = KIND == KIND_R1CS ? 3 : 2, T3 = 0;
  constexpr bool P0 = (PREMASK & 1) != 0, P1 = (PREMASK & 2) != 0, P2 = (PREMASK & 4) != 0, P3 = (PREMASK & 8) != 0;
  const uint32_t* s0 = a.src[T0]; const uint32_t* s1 = a.src[T1];
  const uint32_t* s2 = a.src[NT > 2 ? T2 : T0]; const uint32_t* s3 = a.src[NT > 3 ? T3 : T0];
  uint32_t* d0 = a.dst[T0]; uint32_t* d1 = a.dst[T1];
  uint32_t* d2 = a.dst[NT > 2 ? T2 : T0]; uint32_t* d3 = a.dst[NT > 3 ? T3 : T0];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= q) return;
  ScQuad zn = sc_quad_load<P0>(s0, i, q);
  // the scheduler otherwise moves every prefetch down to its first use (fewer live registers, and the serial chain back):
  // nothing crosses SC_PIN, so the loads stay issued ahead of the arithmetic of the stage they overlap
#define SC_PIN() __builtin_amdgcn_sched_barrier(0)
  for (;;) {
    const size_t inext = i + stride;
    const size_t ipre = inext < q ? inext : i;          // the last iteration re-loads its own first table (harmless) instead of branching
    Fr p0, p2, p3;
    {
      const ScQuad z = zn; zn = sc_quad_load<P1>(s1, i, q); SC_PIN();
      const ScPair t = sc_quad_bind<P0>(z, d0, i, q, r);
      if (KIND == KIND_QUAD) { p0 = t.lo; p2 = fe_sub(fe_dbl(t.hi), t.lo); }
      else { const Fr d = fe_sub(t.hi, t.lo); p0 = t.lo; p2 = fe_add(t.hi, d); p3 = fe_add(p2, d); }      // p(2) = 2 hi - lo, p(3) = p(2) + hi - lo
    }
    {
      const ScQuad z = zn;
      if (NT > 2) zn = sc_quad_load<P2>(s2, i, q); else zn = sc_quad_load<P0>(s0, ipre, q);
      SC_PIN();
      const ScPair t = sc_quad_bind<P1>(z, d1, i, q, r);
      if (KIND == KIND_QUAD) { p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, fe_sub(fe_dbl(t.hi), t.lo)); }
      else { const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d); p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, v); p3 = fe_mul(p3, fe_add(v, d)); }
    }
    if (NT > 2) {
      const ScQuad z = zn;
      if (NT > 3) zn = sc_quad_load<P3>(s3, i, q); else zn = sc_quad_load<P0>(s0, ipre, q);
      SC_PIN();
      const ScPair t = sc_quad_bind<P2>(z, d2, i, q, r);
      const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d);
      if (KIND == KIND_CUBIC) { p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, v); p3 = fe_mul(p3, fe_add(v, d)); }
      else { p0 = fe_sub(p0, t.lo); p2 = fe_sub(p2, v); p3 = fe_sub(p3, fe_add(v, d)); }
    }
    if (NT > 3) {
      const ScQuad z = zn;
      zn = sc_quad_load<P0>(s0, ipre, q); SC_PIN();
      const ScPair t = sc_quad_bind<P3>(z, d3, i, q, r);
      const Fr d = fe_sub(t.hi, t.lo), v = fe_add(t.hi, d);
      p0 = fe_mul(p0, t.lo); p2 = fe_mul(p2, v); p3 = fe_mul(p3, fe_add(v, d));
    }
    e0 = fe_add(e0, p0); e2 = fe_add(e2, p2);
    if (KIND != KIND_QUAD) e3 = fe_add(e3, p3);
    if (inext >= q) break;
    i = inext;
  }
#undef SC_PIN
}
// host side: the masks this kernel is built for (anything else takes the plain kernel)
__host__ __device__ inline bool sc_pf_mask_supported(int kind, unsigned mask) { return mask == 0 || (kind == KIND_CUBIC && mask == 4u); }
template <int KIND>
__global__ void __launch_bounds__(256, 2) k_sc_bind_eval_pf(const ScFusedArgs* __restrict__ args, ScFusedPack pack, size_t q /* old_len / 4 */, ScScalar rmont /* Montgomery form */, uint32_t* __restrict__ partial) {
  ScFusedArgs a;
  if (args) a = args[blockIdx.y];
  else {
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  }
  Fr r; for (int k = 0; k < 8; k++) r.v[k] = rmont.v[k];
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  if (KIND == KIND_CUBIC && a.pre[2]) sc_pf_body<KIND, (KIND == KIND_CUBIC ? 4 : 0)>(a, q, r, e0, e2, e3);   // "par" instance: the shared C arrives bound
  else sc_pf_body<KIND, 0>(a, q, r, e0, e2, e3);
  __shared__ uint32_t sm[4][3][8];
  e0 = wave_sum_fr(e0); e2 = wave_sum_fr(e2);
  if (KIND != KIND_QUAD) e3 = wave_sum_fr(e3);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) { sm[wv][0][k] = e0.v[k]; sm[wv][1][k] = e2.v[k]; sm[wv][2][k] = e3.v[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < 4; w++) { Fr x; for (int k = 0; k < 8; k++) x.v[k] = sm[w][threadIdx.x][k]; s = fe_add(s, x); }
    uint32_t* o = partial + 8 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x);
    for (int k = 0; k < 8; k++) o[k] = s.v[k];
  }
}

// out-of-place bind of the top variable of ONE table: dst[i] = Z[i] + r (Z[i + half] - Z[i]), i < half.  Runs ahead of a fused
// round for a table that several instances of the round share (see ScFusedArgs::pre).
__global__ void __launch_bounds__(256) k_bind_oop(const uint32_t* __restrict__ Z, uint32_t* __restrict__ dst, size_t half, ScScalar rmont) {
  Fr r; for (int k = 0; k < 8; k++) r.v[k] = rmont.v[k];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    const Fr lo = fe_load<FrP>(Z + 8 * i), hi = fe_load<FrP>(Z + 8 * (i + half));
    fe_store<FrP>(dst + 8 * i, fe_add(lo, fe_mul(r, fe_sub(hi, lo))));
  }
}

// out[inst][3] canonical; one 64-lane block per instance folds `nblk` partial triples
__global__ void __launch_bounds__(64) k_sc_finish(const uint32_t* __restrict__ partial, int nblk, uint32_t* __restrict__ out) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  for (int q = 0; q < 3; q++) {
    Fr s = fe_zero<FrP>();
    for (int b = lane; b < nblk; b += 64) { Fr x = fe_load<FrP>(partial + 8 * (((size_t)inst * nblk + b) * 3 + q)); s = fe_add(s, x); }
    s = wave_sum_fr(s);
    if (lane == 0) fe_store<FrP>(out + 8 * ((size_t)inst * 3 + q), fe_from_mont(s));
  }
}

// bind the top variable of up to `count` tables (blockIdx.y = table) to r (Montgomery form, in `rm`)
__global__ void __launch_bounds__(256) k_bind_top(uint32_t* const* __restrict__ tabs, size_t half, const uint32_t* __restrict__ rm) {
  uint32_t* Z = tabs[blockIdx.y];
  const Fr r = fe_load<FrP>(rm);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Fr lo = fe_gload<FrP>(Z + 8 * i), hi = fe_gload<FrP>(Z + 8 * (i + half));
    fe_gstore<FrP>(Z + 8 * i, fe_add(lo, fe_mul(r, fe_sub(hi, lo))));
  }
}

// the same with the table pointers and the challenge as kernel arguments (up to BIND_PACK_MAX tables): nothing to stage, nothing to wait for
constexpr int BIND_PACK_MAX = 64;
struct BindPack { uint32_t* t[BIND_PACK_MAX]; };
__global__ void __launch_bounds__(256) k_bind_top_packed(BindPack pack, size_t half, ScScalar rc) {
  uint32_t* Z = nullptr;
#pragma unroll
  for (int i = 0; i < BIND_PACK_MAX; i++) if (i == (int)blockIdx.y) Z = pack.t[i];
  Fr r; for (int k = 0; k < 8; k++) r.v[k] = rc.v[k];
  r = fe_to_mont(r);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Fr lo = fe_gload<FrP>(Z + 8 * i), hi = fe_gload<FrP>(Z + 8 * (i + half));
    fe_gstore<FrP>(Z + 8 * i, fe_add(lo, fe_mul(r, fe_sub(hi, lo))));
  }
}
// (also run in place: no __restrict__)
__global__ void __launch_bounds__(256) k_fr_to_mont(const uint32_t* in, uint32_t* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fe_store<FrP>(out + 8 * i, fe_to_mont(fe_load<FrP>(in + 8 * i)));
}
__global__ void __launch_bounds__(256) k_fr_from_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fe_store<FrP>(out + 8 * i, fe_from_mont(fe_load<FrP>(in + 8 * i)));
}

// one level of EqPolynomial::evals: out[2k+1] = in[k]*r_j ; out[2k] = in[k] - out[2k+1]   (hyrax.rs:360-366)
__global__ void __launch_bounds__(256) k_eq_level(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t size_in, const uint32_t* __restrict__ rj_mont) {
  const Fr r = fe_load<FrP>(rj_mont);
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < size_in; k += (size_t)gridDim.x * blockDim.x) {
    Fr s = fe_load<FrP>(in + 8 * k);
    Fr hi = fe_mul(s, r);
    fe_store<FrP>(out + 8 * (2 * k + 1), hi);
    fe_store<FrP>(out + 8 * (2 * k), fe_sub(s, hi));
  }
}
// two levels in one pass (r_j then r_{j+1}): in[k] -> out[4k .. 4k+3]; saves the intermediate table's write and re-read
__global__ void __launch_bounds__(256) k_eq_level2(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t size_in, const uint32_t* __restrict__ rj_mont) {
  const Fr r0 = fe_load<FrP>(rj_mont), r1 = fe_load<FrP>(rj_mont + 8);
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < size_in; k += (size_t)gridDim.x * blockDim.x) {
    const Fr s = fe_load<FrP>(in + 8 * k);
    const Fr h = fe_mul(s, r0), l = fe_sub(s, h);
    const Fr hh = fe_mul(h, r1), lh = fe_mul(l, r1);
    fe_store<FrP>(out + 8 * (4 * k), fe_sub(l, lh));
    fe_store<FrP>(out + 8 * (4 * k + 1), lh);
    fe_store<FrP>(out + 8 * (4 * k + 2), fe_sub(h, hh));
    fe_store<FrP>(out + 8 * (4 * k + 3), hh);
  }
}
// the challenge vector as a kernel argument (no staging copy, no host wait): canonical -> Montgomery into r_mont[0..ell)
constexpr int EQ_MAX_VARS = 40;
struct EqPoint { uint32_t v[EQ_MAX_VARS][8]; };
__global__ void __launch_bounds__(64) k_eq_prepare(EqPoint r, int ell, uint32_t* __restrict__ r_mont) {
  const int j = threadIdx.x;
  if (j >= ell) return;
  Fr x;
#pragma unroll 1
  for (int q = 0; q < EQ_MAX_VARS; q++) if (q == j) { for (int k = 0; k < 8; k++) x.v[k] = r.v[q][k]; }
  fe_store<FrP>(r_mont + 8 * j, fe_to_mont(x));
}
// the first m levels at once: out[i] = prod_j (bit_{m-1-j}(i) ? r_j : 1 - r_j), i < 2^m — the same field element the level-by-level
// recurrence s -> (s - s r, s r) of hyrax.rs:360-366 produces (variable 0 is the most significant index bit)
__global__ void __launch_bounds__(256) k_eq_direct(const uint32_t* __restrict__ r_mont, int m, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((size_t)1 << m)) return;
  const Fr one = fe_one<FrP>();
  Fr acc = one;
  for (int j = 0; j < m; j++) {
    const Fr r = fe_load<FrP>(r_mont + 8 * j);
    acc = fe_mul(acc, ((i >> (m - 1 - j)) & 1) ? r : fe_sub(one, r));
  }
  fe_store<FrP>(out + 8 * i, acc);
}
// hash layer: out[j] = (ts[j]+ts_add)*g^2 + val[j]*g + addr[j] - tau.  g2rr = g^2 * R (a "doubly Montgomery" value), so one
// Montgomery product with the plain small integer ts gives mont(ts*g^2); rr = R^2 turns the plain addr into mont(addr).
__global__ void __launch_bounds__(256) k_hash_layer(const uint32_t* __restrict__ addr, const uint32_t* __restrict__ val, const uint32_t* __restrict__ ts, uint32_t ts_add,
                                                    const uint32_t* __restrict__ consts /* g, g2rr, tau : Montgomery */, size_t n, uint32_t* __restrict__ out) {
  const Fr g = fe_load<FrP>(consts), g2rr = fe_load<FrP>(consts + 8), tau = fe_load<FrP>(consts + 16);
  Fr rr; for (int k = 0; k < 8; k++) rr.v[k] = FrP::R2[k];
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    Fr a = fe_zero<FrP>(), t = fe_zero<FrP>();
    const unsigned long long av = addr ? addr[j] : (unsigned long long)j;
    a.v[0] = (uint32_t)av; a.v[1] = (uint32_t)(av >> 32);
    const unsigned long long tv = (unsigned long long)(ts ? ts[j] : 0u) + ts_add;
    t.v[0] = (uint32_t)tv; t.v[1] = (uint32_t)(tv >> 32);
    Fr acc = fe_mul(g2rr, t);
    acc = fe_add(acc, fe_mul(fe_load<FrP>(val + 8 * j), g));
    acc = fe_add(acc, fe_mul(rr, a));
    fe_store<FrP>(out + 8 * j, fe_sub(acc, tau));
  }
}
__global__ void __launch_bounds__(256) k_product_layer(const uint32_t* __restrict__ in, size_t half, uint32_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
    fe_store<FrP>(out + 8 * i, fe_mul(fe_load<FrP>(in + 8 * i), fe_load<FrP>(in + 8 * (i + half))));
}
// g (canonical) -> {mont(g), g^2 * R^2 mod r (= mont(mont(g^2))), mont(tau)}
// the top of a product tree in one launch: from a layer of `len` <= 2048 entries down to the single product, one block, one layer
// after the other (out[k] has len >> (k+1) entries); saves ~10 launches per circuit, which is what the small layers cost
constexpr int PT_TAIL_MAX = 12;
struct ProductTail { uint32_t* out[PT_TAIL_MAX]; };
__global__ void __launch_bounds__(1024) k_product_tail(const uint32_t* __restrict__ in, size_t len, ProductTail pt) {
  const uint32_t* src = in;
  int k = 0;
  for (size_t half = len / 2; half >= 1; half >>= 1, k++) {
    uint32_t* dst = nullptr;
#pragma unroll
    for (int i = 0; i < PT_TAIL_MAX; i++) if (i == k) dst = pt.out[i];
    for (size_t i = threadIdx.x; i < half; i += blockDim.x) fe_store<FrP>(dst + 8 * i, fe_mul(fe_load<FrP>(src + 8 * i), fe_load<FrP>(src + 8 * (i + half))));
    __threadfence_block();
    __syncthreads();
    src = dst;
    if (half == 1) break;
  }
}
__global__ void k_hash_consts(ScScalar g_canon, ScScalar tau_canon, uint32_t* __restrict__ consts) {
  if (threadIdx.x || blockIdx.x) return;
  Fr g, t; for (int k = 0; k < 8; k++) { g.v[k] = g_canon.v[k]; t.v[k] = tau_canon.v[k]; }
  const Fr gm = fe_to_mont(g), tm = fe_to_mont(t);
  fe_store<FrP>(consts, gm);
  fe_store<FrP>(consts + 8, fe_to_mont(fe_mul(gm, gm)));
  fe_store<FrP>(consts + 16, tm);
}

// <a, b> partial sums: partial[block] (Montgomery); finished by k_sc_finish-style fold on one value
__global__ void __launch_bounds__(256) k_dot(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, uint32_t* __restrict__ partial) {
  Fr acc = fe_zero<FrP>();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc = fe_add(acc, fe_mul(fe_load<FrP>(a + 8 * i), fe_load<FrP>(b + 8 * i)));
  __shared__ uint32_t sm[4][8];
  acc = wave_sum_fr(acc);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) for (int k = 0; k < 8; k++) sm[wv][k] = acc.v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < 4; w++) { Fr x; for (int k = 0; k < 8; k++) x.v[k] = sm[w][k]; s = fe_add(s, x); }
    // stored in the 3-slot layout of k_sc_finish (slots 1, 2 zero)
    uint32_t* o = partial + 8 * ((size_t)blockIdx.x * 3);
    for (int k = 0; k < 8; k++) { o[k] = s.v[k]; o[8 + k] = 0; o[16 + k] = 0; }
  }
}
// <Z_i, chi> for up to SC_PACK_MAX tables at once (blockIdx.y = table): the evaluations of many polynomials at one point
__global__ void __launch_bounds__(256) k_dot_many(ScArgsPack pack, const uint32_t* __restrict__ chi, size_t n, uint32_t* __restrict__ partial) {
  ScArgs a;
#pragma unroll
  for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  const uint32_t* z = a.t[0];
  Fr acc = fe_zero<FrP>();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc = fe_add(acc, fe_mul(fe_load<FrP>(z + 8 * i), fe_load<FrP>(chi + 8 * i)));
  __shared__ uint32_t sm[4][8];
  acc = wave_sum_fr(acc);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) for (int k = 0; k < 8; k++) sm[wv][k] = acc.v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < 4; w++) { Fr x; for (int k = 0; k < 8; k++) x.v[k] = sm[w][k]; s = fe_add(s, x); }
    uint32_t* o = partial + 8 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3);
    for (int k = 0; k < 8; k++) { o[k] = s.v[k]; o[8 + k] = 0; o[16 + k] = 0; }
  }
}
// L*Z: block (column tile of 64, row slice): lanes own columns (coalesced 2 KiB rows), the block's 4 waves split the row
// slice; partial[slice][col] then folded by k_bound_fold.  Z is L_size x R_size row-major.
__global__ void __launch_bounds__(256) k_bound_partial(const uint32_t* __restrict__ Z, const uint32_t* __restrict__ Lv, size_t L_size, size_t R_size, size_t rows_per_slice, uint32_t* __restrict__ partial) {
  const size_t col = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int wv = threadIdx.x >> 6;
  const size_t j0 = (size_t)blockIdx.y * rows_per_slice, j1 = (j0 + rows_per_slice < L_size) ? j0 + rows_per_slice : L_size;
  Fr acc = fe_zero<FrP>();
  if (col < R_size)
    for (size_t j = j0 + wv; j < j1; j += 4) acc = fe_add(acc, fe_mul(fe_load<FrP>(Lv + 8 * j), fe_load<FrP>(Z + 8 * (j * R_size + col))));
  __shared__ uint32_t sm[4][64][8];
  for (int k = 0; k < 8; k++) sm[wv][threadIdx.x & 63][k] = acc.v[k];
  __syncthreads();
  if (wv == 0 && col < R_size) {
    Fr s = acc;
    for (int w = 1; w < 4; w++) { Fr x; for (int k = 0; k < 8; k++) x.v[k] = sm[w][threadIdx.x & 63][k]; s = fe_add(s, x); }
    fe_store<FrP>(partial + 8 * ((size_t)blockIdx.y * R_size + col), s);
  }
}
__global__ void __launch_bounds__(256) k_bound_fold(const uint32_t* __restrict__ partial, size_t nslices, size_t R_size, uint32_t* __restrict__ out) {
  const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= R_size) return;
  Fr s = fe_zero<FrP>();
  for (size_t k = 0; k < nslices; k++) s = fe_add(s, fe_load<FrP>(partial + 8 * (k * R_size + col)));
  fe_store<FrP>(out + 8 * col, s);
}

// out[k*n + i] = mem[k][addr[k][i]]  (AddrTimestamps::deref_mem, sparse_mlpoly_full.rs:245-252), zero padding past count*n
struct GatherArgs { const uint32_t* mem; const uint32_t* addr; size_t mem_len; };
__global__ void __launch_bounds__(256) k_gather_merge(const GatherArgs* __restrict__ args, size_t count, size_t n, size_t padded, uint32_t* __restrict__ out, uint32_t* __restrict__ oob) {
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < padded; t += (size_t)gridDim.x * blockDim.x) {
    Fr v = fe_zero<FrP>();
    if (t < count * n) {
      const size_t k = t / n, i = t - k * n;
      const GatherArgs a = args[k];
      const uint32_t idx = a.addr[i];
      if (idx < a.mem_len) v = fe_load<FrP>(a.mem + 8 * (size_t)idx); else atomicAdd(oob, 1u);   // sparse_mlpoly_full.rs:228 assert!(addr < num_cells)
    }
    fe_store<FrP>(out + 8 * t, v);
  }
}
__global__ void k_fr_set_one(uint32_t* out) { if (threadIdx.x == 0 && blockIdx.x == 0) fe_store<FrP>(out, fe_one<FrP>()); }

}  // namespace sbn
