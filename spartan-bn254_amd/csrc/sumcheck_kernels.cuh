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
// Tables are AoS arrays of 32-byte Montgomery-form elements (fp.cuh's memory format: any non-negative representative below
// 2.5 r; only what leaves the library is canonical); lane i reads element i with two 16-byte loads, so a wavefront reads 2 KiB
// contiguous per table half — the kernels are HBM-streaming with ~1 product per 32 B.
// Field addition is associative and commutative mod r, so the tree-shaped partial sums give the same canonical
// result as the reference's left-to-right loop.
#pragma once
#include "fp.cuh"

namespace sbn {

enum { KIND_CUBIC = 0, KIND_R1CS = 1, KIND_QUAD = 2 };

struct ScArgs {          // one instance: up to 4 tables, all of length 2*half
  const uint32_t* t[4];
};

// lane 0 <- the sum over the wavefront (normalised; the value grows 64-fold: callers reduce first)
__device__ __forceinline__ Fr wave_sum_fr(Fr v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fe_add(v, fe_shfl_down(v, d));
  return v;
}
// running sums of products: acc += x (normalised), with a Montgomery reduction every 64 terms so that the lazy value stays far
// inside the top limb's range whatever the trip count
__device__ __forceinline__ void fr_acc(Fr& acc, const Fr& x, uint32_t& cnt) {
  acc = fe_add(acc, x);
  if ((++cnt & 63u) == 0) acc = fe_reduce(acc);
}
// the same for terms up to 4 r (fe_reduce takes values below 169 r): a reduction every 32 terms
__device__ __forceinline__ void fr_acc32(Fr& acc, const Fr& x, uint32_t& cnt) {
  acc = fe_add(acc, x);
  if ((++cnt & 31u) == 0) acc = fe_reduce(acc);
}

// ---- single-launch rounds -------------------------------------------------------------------------------------------
// A sumcheck round is a host round trip (the next challenge comes from the transcript), and below ~2^16 entries the trip,
// not the data, is the cost.  So a round is ONE launch: the table pointers and the challenge travel as kernel arguments (no
// staging copy), and the last block of every instance to finish — found with a ticket counter — folds the per-block partial
// sums and stores the canonical values straight into host-visible pinned memory (no finishing kernel, no copy back).
constexpr int SC_PACK_MAX = 24;            // instances whose pointers fit in the kernel-argument block
struct ScScalar { uint32_t v[8]; };        // a scalar passed by value: 8 x 32-bit words (memory format)
__device__ __forceinline__ Fr fr_from_words(const ScScalar& s) { return fe_unpack<FrP>(s.v); }

__device__ __forceinline__ Fr fr_load_coherent(const uint32_t* p) {   // written by other blocks of this launch: bypass the CU's L1
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return fe_unpack<FrP>(w);
}
// partial sums that another block of the launch will read: written THROUGH the caches (system-scope atomic stores), so that the
// hand-over needs no device-scope release fence — on gfx950 that fence writes back every dirty line of the XCD's L2, i.e. the
// megabytes of freshly bound table values next to these 96 bytes (measured: the round on 2^15-entry tables 95 -> see DESIGN 4.2)
__device__ __forceinline__ void fr_store_coherent(uint32_t* p, const Fr& a) {
  uint32_t w[8]; fe_pack<FrP>(fe_fix_tab<FrP>(fe_norm(a)), w);
#pragma unroll
  for (int k = 0; k < 8; k++) __hip_atomic_store(p + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The storing wave waits for the write acknowledgements of everything it has stored.  Every wave that stores bytes another block (or
// the host) will read runs this BEFORE the workgroup barrier that precedes the signal (ticket add / mailbox flag): a barrier does not
// drain vector memory, and a workgroup-scope fence compiles to nothing on gfx950 outside threadgroup-split mode, so without this wait
// the signal can overtake the stores (MI355X_MICROARCH.md, "Valid forms": sc1 stores -> every storing wave's vmcnt(0) -> barrier ->
// one lane signals).  Inline asm on purpose: the compiler's own waitcnt insertion does not see a cross-wave dependency, and
// tests/test_isa_handover.py checks the shipped code object for this instruction between the stores and the signal.
__device__ __forceinline__ void sc_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// mailbox flag: results first (drained by their own waves, then the barrier), then a system-scope release, its write-back waited for
// explicitly (the compiler may drop the wait behind buffer_wbl2 when it believes the wave has nothing outstanding), then the flag.
__device__ __forceinline__ void sc_flag_store(uint32_t* flag, uint32_t seq) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int SC_MBOX_FLAGS = SC_PACK_MAX * 24;      // word offset of the flags inside the mailbox (SC_PACK_MAX round flags + 1 for the final claims)
constexpr int SC_MBOX_FINALS = SC_MBOX_FLAGS + SC_PACK_MAX + 8;      // word offset of the final claims of a stateful sumcheck (up to 3 * SC_PACK_MAX + 1 tables x 8 words)
constexpr int SC_MBOX_WORDS = SC_MBOX_FINALS + (3 * SC_PACK_MAX + 8) * 8;
// The block's three running sums -> partial[(inst * gridDim.x + blockIdx.x) * 3 + q] (memory format, Montgomery domain).
// mbox != nullptr with ONE block per instance (the tables of the last ~8 rounds): the block's sums are the round's values, so
// they go straight to the host mailbox as canonical integers and the flag follows — no partial sums, no ticket, no device fence.
// Returns true when it finished the round that way.
// slot = the "instance" index of the partial-sum / ticket / mailbox arrays (blockIdx.y for the per-instance kernels)
// bidx / nblk: this block's index among the blocks that work for `slot`, and their number (blockIdx.x / gridDim.x in the one-kind kernels)
__device__ __forceinline__ bool sc_block_sums_store_slot(Fr e0, Fr e2, Fr e3, bool three, uint32_t* __restrict__ partial, uint32_t* __restrict__ mbox, uint32_t seq, uint32_t slot, uint32_t bidx, uint32_t nblk) {
  __shared__ uint32_t sm[4][3][NL];
  e0 = wave_sum_fr(fe_reduce(e0)); e2 = wave_sum_fr(fe_reduce(e2));
  if (three) e3 = wave_sum_fr(fe_reduce(e3));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NL; k++) { sm[wv][0][k] = e0.v[k]; sm[wv][1][k] = e2.v[k]; sm[wv][2][k] = e3.v[k]; }
  }
  __syncthreads();
  const bool direct = mbox != nullptr && nblk == 1;
  if (threadIdx.x < 3) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < 4; w++) { Fr x; for (int k = 0; k < NL; k++) x.v[k] = sm[w][threadIdx.x][k]; s = fe_add(s, x); }
    if (direct) fe_store_packed<FrP>(mbox + 8 * ((size_t)slot * 3 + threadIdx.x), fe_from_mont(s));
    else fr_store_coherent(partial + 8 * (((size_t)slot * nblk + bidx) * 3 + threadIdx.x), fe_reduce(s));
  }
  sc_drain_stores();                           // (wave 0 stored; the others have nothing outstanding) — before ANY barrier that precedes a signal
  if (direct) {
    __syncthreads();                           // the three stores have been acknowledged: the flag may follow
    if (threadIdx.x == 0) sc_flag_store(mbox + SC_MBOX_FLAGS + slot, seq);
  }
  return direct;
}
__device__ __forceinline__ bool sc_block_sums_store(Fr e0, Fr e2, Fr e3, bool three, uint32_t* __restrict__ partial, uint32_t* __restrict__ mbox, uint32_t seq) {
  return sc_block_sums_store_slot(e0, e2, e3, three, partial, mbox, seq, blockIdx.y, blockIdx.x, gridDim.x);
}
__device__ __forceinline__ void sc_block_sums_store_at(Fr e0, Fr e2, Fr e3, uint32_t* __restrict__ partial, uint32_t slot, uint32_t bidx, uint32_t nblk) {
  sc_block_sums_store_slot(e0, e2, e3, true, partial, nullptr, 0u, slot, bidx, nblk);
}
// Called by all 256 threads after the block's triple went to partial[...] (several blocks per instance).  nq = 2 or 3.
// `out` = mailbox in coherent pinned host memory: 24 x 96 B of results, then one flag word per instance; the flag is stored
// (release, system scope) after the results, so the host can poll it instead of paying a stream synchronisation.
__device__ __forceinline__ void sc_last_block_fold_at(const uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, int nq, uint32_t seq, uint32_t slot, uint32_t nblk) {
  __shared__ uint32_t s_last;
  // The block's triple was written through to memory (fr_store_coherent) and its wave has waited for the acknowledgements
  // (sc_drain_stores at the end of sc_block_sums_store); behind the barrier one lane takes the ticket.  No L2 write-back anywhere.
  __syncthreads();
  if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(&tickets[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1) ? 1u : 0u;
  __syncthreads();                             // the add has returned (its value was used) before any wave of the last block loads
  if (!s_last) return;
  // the partial sums are read with system-scope loads straight to registers (they bypass this CU's L1); the agent-scope acquire on
  // top is what the hand-over rules ask for outside the measured one-workgroup-per-CU case: it invalidates this CU's L1 only
  if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wv < 3) {
    Fr s = fe_zero<FrP>();
    if (wv < nq) for (unsigned b = lane; b < nblk; b += 64) s = fe_add(s, fr_load_coherent(partial + 8 * (((size_t)slot * nblk + b) * 3 + wv)));
    s = wave_sum_fr(fe_reduce(s));
    if (lane == 0) fe_store_packed<FrP>(out + 8 * ((size_t)slot * 3 + wv), fe_from_mont(s));
  }
  sc_drain_stores();                           // waves 0..2 each wait for their own result store
  __syncthreads();                             // ... and only then does wave 0 raise the flag
  if (threadIdx.x == 0) {
    __hip_atomic_store(&tickets[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (written through: the next launch's adds come from any XCD)
    sc_flag_store(out + SC_MBOX_FLAGS + slot, seq);
  }
}
__device__ __forceinline__ void sc_last_block_fold(const uint32_t* __restrict__ partial, uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, int nq, uint32_t seq) {
  sc_last_block_fold_at(partial, tickets, out, nq, seq, blockIdx.y, gridDim.x);
}

// The values of one table's line lo + t (hi - lo) at t = 2 and 3: p(2) = 2 hi - lo, p(3) = p(2) + hi - lo  (sumcheck.rs:111-135),
// normalised (products take them as operands); lo, hi: normalised table values
struct ScPts { Fr v2, v3; };
__device__ __forceinline__ ScPts sc_points(const Fr& lo, const Fr& hi) {
  const Fr d = fe_sub_lazy(hi, lo);
  ScPts o; o.v2 = fe_norm(fe_add_lazy(hi, d)); o.v3 = fe_norm(fe_add_lazy(o.v2, d));
  return o;
}
// The same on fp.cuh's unsigned fast path, for table values in [0, 2.5 r) (what this library's sumcheck tables hold: fe_fix_tab):
// the difference carries 3r, so v2 = 2 hi - lo + 3r < 8r and v3 = 3 hi - 2 lo + 6r < 13.5r, all limbs non-negative
// (13.5r * 13.5r = 182 r^2 / 2^261 = 1.08 r: a product of two such points comes back below 2.1 r).
__device__ __forceinline__ ScPts sc_points_u(const Fr& lo, const Fr& hi) {
  const Fr d = fe_subb<FrP, 3, 1>(hi, lo);
  const Fr v2 = fe_add_lazy(hi, d);
  ScPts o; o.v2 = fe_normu(v2); o.v3 = fe_normu(fe_add_lazy(v2, d));
  return o;
}

__device__ __forceinline__ Fr sc_point2_u(const Fr& lo, const Fr& hi) { return fe_normu(fe_add_lazy(hi, fe_subb<FrP, 3, 1>(hi, lo))); }   // 2 hi - lo + 3 r

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
  uint32_t c0 = 0, c2 = 0, c3 = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    if (KIND == KIND_QUAD) {
      const Fr zl = fe_gload<FrP>(a.t[0] + 8 * i), zh = fe_gload<FrP>(a.t[0] + 8 * (i + half));
      const Fr al = fe_gload<FrP>(a.t[1] + 8 * i), ah = fe_gload<FrP>(a.t[1] + 8 * (i + half));
      fr_acc(e0, fe_mulu(zl, al), c0);
      fr_acc(e2, fe_mulu(sc_point2_u(zl, zh), sc_point2_u(al, ah)), c2);
    } else if (KIND == KIND_CUBIC) {
      const Fr al = fe_gload<FrP>(a.t[0] + 8 * i), ah = fe_gload<FrP>(a.t[0] + 8 * (i + half));
      const Fr bl = fe_gload<FrP>(a.t[1] + 8 * i), bh = fe_gload<FrP>(a.t[1] + 8 * (i + half));
      const Fr cl = fe_gload<FrP>(a.t[2] + 8 * i), ch = fe_gload<FrP>(a.t[2] + 8 * (i + half));
      const ScPts pa = sc_points_u(al, ah), pb = sc_points_u(bl, bh), pc = sc_points_u(cl, ch);
      fr_acc(e0, fe_mulu(fe_mulu(al, bl), cl), c0);
      fr_acc(e2, fe_mulu(fe_mulu(pa.v2, pb.v2), pc.v2), c2);
      fr_acc(e3, fe_mulu(fe_mulu(pa.v3, pb.v3), pc.v3), c3);
    } else {
      const Fr tl = fe_gload<FrP>(a.t[0] + 8 * i), th = fe_gload<FrP>(a.t[0] + 8 * (i + half));
      const Fr al = fe_gload<FrP>(a.t[1] + 8 * i), ah = fe_gload<FrP>(a.t[1] + 8 * (i + half));
      const Fr bl = fe_gload<FrP>(a.t[2] + 8 * i), bh = fe_gload<FrP>(a.t[2] + 8 * (i + half));
      const Fr cl = fe_gload<FrP>(a.t[3] + 8 * i), ch = fe_gload<FrP>(a.t[3] + 8 * (i + half));
      // T (A B - C) on the unsigned fast path: the subtrahends are table points below 2.5 r, 8 r, 13.5 r (sc_points_u), so the
      // differences carry 3 r, 9 r, 14 r and stay non-negative; the outer products (< 15.1 r * 13.5 r) return values below 2.3 r
      const ScPts pt = sc_points_u(tl, th), pa = sc_points_u(al, ah), pb = sc_points_u(bl, bh), pc = sc_points_u(cl, ch);
      fr_acc(e0, fe_mulu(tl, fe_subb<FrP, 3, 1>(fe_mulu(al, bl), cl)), c0);
      fr_acc(e2, fe_mulu(pt.v2, fe_subb<FrP, 9, 1>(fe_mulu(pa.v2, pb.v2), pc.v2)), c2);
      fr_acc(e3, fe_mulu(pt.v3, fe_subb<FrP, 14, 1>(fe_mulu(pa.v3, pb.v3), pc.v3)), c3);
    }
  }
  if (sc_block_sums_store(e0, e2, e3, KIND != KIND_QUAD, partial, tickets ? out : nullptr, seq)) return;
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
// The challenge arrives in Montgomery form (one host product instead of one per thread), and the grid is sized for a few block
// rounds so that the wave/LDS reduction at the end is amortised over many indices per thread.
struct ScFusedArgs {
  const uint32_t* src[4];
  uint32_t* dst[4];
  uint32_t pre[4];
};
struct ScFusedPack { ScFusedArgs a[SC_PACK_MAX]; };

// z0 + r (z2 - z0) as the table representative in [0, 2.5 r) that is stored and used: z0, z2 are table values (normalised,
// below 2.5 r), their limb-wise difference is a legal product operand, the product lies in (-0.1 r, 1.1 r)
// On the unsigned fast path: table values lie in [0, 2.5 r) (normalised), the difference carries 3r (below 5.5 r), the product of the
// canonical challenge with it lies in [0, 1.1 r), and the sum (< 3.6 r) drops 2r when its top limb says it reached 2r — decided on
// the un-carried top limb (the lower limbs add at most one unit to it), so one carry pass serves the sum and the correction.
__device__ __forceinline__ Fr sc_bind1(const Fr& z0, const Fr& z2, const Fr& r) {
  Fr x = fe_add_lazy(z0, fe_mulu(r, fe_subb<FrP, 3, 1>(z2, z0)));
  constexpr int32_t P8 = (int32_t)FrP::P29[8];
  const uint32_t big = (uint32_t)((2 * P8 + 1 - (int32_t)x.v[NL - 1]) >> 31);          // top >= 2 P8 + 2: value > 2r
#pragma unroll
  for (int k = 0; k < NL; k++) x.v[k] -= kp29<FrP, 2>(k) & big;
  return fe_norm(x);                                                                    // (limbs may have dipped below zero: signed carries)
}
// one table at index i: bound values lo = Z'[i], hi = Z'[i + q]
struct ScPair { Fr lo, hi; };
__device__ __forceinline__ ScPair sc_bound_pair(const uint32_t* __restrict__ z, uint32_t* __restrict__ dst, uint32_t pre, size_t i, size_t q, const Fr& r) {
  ScPair o;
  if (pre) { o.lo = fe_gload<FrP>(z + 8 * i); o.hi = fe_gload<FrP>(z + 8 * (i + q)); return o; }
  const Fr z0 = fe_gload<FrP>(z + 8 * i), z2 = fe_gload<FrP>(z + 8 * (i + 2 * q));
  const Fr z1 = fe_gload<FrP>(z + 8 * (i + q)), z3 = fe_gload<FrP>(z + 8 * (i + 3 * q));
  o.lo = sc_bind1(z0, z2, r);
  o.hi = sc_bind1(z1, z3, r);
  if (dst) { fe_gstore_packed<FrP>(dst + 8 * i, o.lo); fe_gstore_packed<FrP>(dst + 8 * (i + q), o.hi); }
  return o;
}
// the running products of one index at the points 0, 2, 3 and how a table enters them
struct ScProd { Fr p0, p2, p3; };
// (all three comb functions run on fp.cuh's unsigned fast path: table values lie in [0, 2.5 r), their points below 8 r and 13.5 r)
template <int KIND> __device__ __forceinline__ void sc_first(ScProd& P, const ScPair& t) {
  P.p0 = t.lo;
  if (KIND == KIND_QUAD) { P.p2 = sc_point2_u(t.lo, t.hi); return; }
  const ScPts v = sc_points_u(t.lo, t.hi); P.p2 = v.v2; P.p3 = v.v3;
}
template <int KIND> __device__ __forceinline__ void sc_times(ScProd& P, const ScPair& t) {
  P.p0 = fe_mulu(P.p0, t.lo);
  if (KIND == KIND_QUAD) { P.p2 = fe_mulu(P.p2, sc_point2_u(t.lo, t.hi)); return; }
  const ScPts v = sc_points_u(t.lo, t.hi);
  P.p2 = fe_mulu(P.p2, v.v2); P.p3 = fe_mulu(P.p3, v.v3);
}
__device__ __forceinline__ void sc_minus(ScProd& P, const ScPair& t) {       // r1cs: Az*Bz - Cz (+ 3 r, 9 r, 14 r: left un-carried, the next step multiplies by normalised points of tau)
  const ScPts v = sc_points_u(t.lo, t.hi);
  P.p0 = fe_subb<FrP, 3, 1>(P.p0, t.lo); P.p2 = fe_subb<FrP, 9, 1>(P.p2, v.v2); P.p3 = fe_subb<FrP, 14, 1>(P.p3, v.v3);
}

// WPS = waves per SIMD the register allocation is held to
template <int KIND, int WPS>
__global__ void __launch_bounds__(256, WPS) k_sc_bind_eval(const ScFusedArgs* __restrict__ args, ScFusedPack pack, size_t q /* old_len / 4 */, ScScalar rmont /* Montgomery form */, uint32_t* __restrict__ partial,
                                                      uint32_t* __restrict__ tickets, uint32_t* __restrict__ out, uint32_t seq) {
  ScFusedArgs a;
  if (args) a = args[blockIdx.y];
  else {
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  }
  const Fr r = fr_from_words(rmont);
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  uint32_t c0 = 0, c2 = 0, c3 = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
#define SC_TAB(j) sc_bound_pair(a.src[j], a.dst[j], a.pre[j], i, q, r)
    ScProd P;
    if (KIND == KIND_QUAD) {
      { const ScPair t = SC_TAB(0); sc_first<KIND>(P, t); }
      { const ScPair t = SC_TAB(1); sc_times<KIND>(P, t); }
    } else if (KIND == KIND_CUBIC) {
      { const ScPair t = SC_TAB(0); sc_first<KIND>(P, t); }
      { const ScPair t = SC_TAB(1); sc_times<KIND>(P, t); }
      { const ScPair t = SC_TAB(2); sc_times<KIND>(P, t); }
    } else {
      // tau * (Az * Bz - Cz): tables 1, 2 (product), 3 (subtracted), then 0 (tau)
      { const ScPair t = SC_TAB(1); sc_first<KIND>(P, t); }
      { const ScPair t = SC_TAB(2); sc_times<KIND>(P, t); }
      { const ScPair t = SC_TAB(3); sc_minus(P, t); }
      { const ScPair t = SC_TAB(0); sc_times<KIND>(P, t); }
    }
#undef SC_TAB
    fr_acc(e0, P.p0, c0); fr_acc(e2, P.p2, c2);
    if (KIND != KIND_QUAD) fr_acc(e3, P.p3, c3);
  }
  if (sc_block_sums_store(e0, e2, e3, KIND != KIND_QUAD, partial, tickets ? out : nullptr, seq)) return;
  if (tickets) sc_last_block_fold(partial, tickets, out, KIND == KIND_QUAD ? 2 : 3, seq);
}

// The fused round on the LAST tables of a sumcheck (q = old_len / 4 <= 128 indices): four lanes per index, one block per instance.  A round on such a table
// is a chain of dependent products on one wave (12 for the cubic comb function: 0.54 us each on a lone wave) followed by three wave
// sums; here the 2 NT binds of an index are dealt over its four lanes (lane `role` binds slot role, then slot 4 + role), the
// bound values are exchanged inside the quad by DPP, and lanes 0, 1, 2 evaluate the comb function at the points 0, 2, 3 — a chain
// of 2 binds + 2 products (3 + 2 for R1CS, 1 + 1 quadratic) — and ONE 4-level butterfly over the quads sums all three points at
// once.  The results go straight to the host mailbox (as sc_block_sums_store's direct path).  Only for the latency-bound rounds:
// all four lanes run the whole instruction stream, 2.4x the work per index of the lane-per-index kernel (measured: slower from
// 512 indices on, and with several blocks per instance the ticketed fold costs 7 us: 16.8 us at 128 indices against 9.6 at 64).
// Slot s = 2 * t + h: table t in the kind's order (cubic 0,1,2 - r1cs 1,2,3,0 - quad 0,1), half h (0: lo = Z'[i], 1: hi = Z'[i + q]).
template <int KIND>
__global__ void __launch_bounds__(512) k_sc_bind_eval_tiny(const ScFusedArgs* __restrict__ args, ScFusedPack pack, size_t q, ScScalar rmont, uint32_t* __restrict__ mbox, uint32_t seq) {
  constexpr int NT = KIND == KIND_QUAD ? 2 : KIND == KIND_CUBIC ? 3 : 4;
  ScFusedArgs a;
  if (args) a = args[blockIdx.y];
  else {
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  }
  const Fr r = fr_from_words(rmont);
  const int lane = threadIdx.x, role = lane & 3;
  const size_t i = (size_t)(lane >> 2);
  const bool live = i < q;                                   // whole quads are live or not
  Fr v[2];                                                   // the lane's bound values: slot role, slot 4 + role
#pragma unroll
  for (int rnd = 0; rnd < 2; rnd++) {
    v[rnd] = fe_zero<FrP>();
    const int slot = 4 * rnd + role;
    if (slot < 2 * NT && live) {
      const int tk = slot >> 1, h = slot & 1;
      int t = 0;                                             // table index of position tk in the kind's order
      if (KIND == KIND_R1CS) t = tk == 3 ? 0 : tk + 1; else t = tk;
      const uint32_t* src = nullptr; uint32_t* dst = nullptr; uint32_t pre = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) if (j == t) { src = a.src[j]; dst = a.dst[j]; pre = a.pre[j]; }
      const size_t e = i + (size_t)h * q;
      if (pre) v[rnd] = fe_gload<FrP>(src + 8 * e);
      else {
        v[rnd] = sc_bind1(fe_gload<FrP>(src + 8 * e), fe_gload<FrP>(src + 8 * (e + 2 * q)), r);
        if (dst) fe_gstore_packed<FrP>(dst + 8 * e, v[rnd]);
      }
    }
  }
  // every lane of the quad gets the (lo, hi) pairs of the tables in the kind's order
  const Fr l0 = fe_quad_bcast<0>(v[0]), h0 = fe_quad_bcast<1>(v[0]), l1 = fe_quad_bcast<2>(v[0]), h1 = fe_quad_bcast<3>(v[0]);
  const Fr l2 = fe_quad_bcast<0>(v[1]), h2 = fe_quad_bcast<1>(v[1]), l3 = fe_quad_bcast<2>(v[1]), h3 = fe_quad_bcast<3>(v[1]);
  // lane `role` takes the point role -> 0, 2, 3 of every table (role 3 idles)
  auto at = [&](const Fr& lo, const Fr& hi) {
    const ScPts p = sc_points_u(lo, hi);
    return fe_sel4(role, lo, p.v2, p.v3, lo);
  };
  Fr e;
  if (KIND == KIND_QUAD) e = fe_mulu(at(l0, h0), at(l1, h1));
  else if (KIND == KIND_CUBIC) e = fe_mulu(fe_mulu(at(l0, h0), at(l1, h1)), at(l2, h2));
  else {
    // tau (A B - C): the subtrahend's cover depends on the point (3 r, 9 r, 14 r: see k_sc_eval); one cover for all lanes: 14 r
    const Fr ab = fe_mulu(at(l0, h0), at(l1, h1));
    e = fe_mulu(at(l3, h3), fe_subb<FrP, 14, 1>(ab, at(l2, h2)));
  }
  const bool counts = live && role < (KIND == KIND_QUAD ? 2 : 3);
  if (!counts) e = fe_zero<FrP>();
  e = fe_reduce(e);
  // sum over the quads of the wave (same role), then over the block's waves
#pragma unroll
  for (int d = 4; d < 64; d <<= 1) e = fe_add(e, fe_shfl_xor(e, d));
  __shared__ uint32_t sm[8][4][NL];
  const int wv = lane >> 6, nw = (int)(blockDim.x >> 6);
  if ((lane & 63) < 4) { for (int k = 0; k < NL; k++) sm[wv][role][k] = e.v[k]; }
  __syncthreads();
  if (lane < (KIND == KIND_QUAD ? 2 : 3)) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < nw; w++) { Fr x; for (int k = 0; k < NL; k++) x.v[k] = sm[w][lane][k]; s = fe_add(s, x); }
    fe_store_packed<FrP>(mbox + 8 * ((size_t)blockIdx.y * 3 + lane), fe_from_mont(s));
  }
  sc_drain_stores();                         // the storing wave waits for its acknowledgements before the barrier
  __syncthreads();
  if (lane == 0) sc_flag_store(mbox + SC_MBOX_FLAGS + blockIdx.y, seq);
}

// The streaming form of the fused round (tables of 2^16 entries and more): the NEXT table's elements are loaded into a second
// register set before the current table's arithmetic starts (software pipeline over the flattened (index, table) sequence), so
// every wave has 8 KB in flight all the time.  rocprofv3 on the plain form: SQ_WAIT_ANY (memory) drops from 21 % to 5 % of the
// wave cycles with it; what remains is VALU issue (two waves per SIMD each ~55 % active = the VALU saturated).
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
  o.lo = sc_bind1(z.z0, z.z2, r);
  o.hi = sc_bind1(z.z1, z.z3, r);
  fe_gstore_packed<FrP>(dst + 8 * i, o.lo); fe_gstore_packed<FrP>(dst + 8 * (i + q), o.hi);
  return o;
}
// table order: cubic 0,1,2 — r1cs 1,2,3,0 (tau last: tau * (Az*Bz - Cz)) — quad 0,1.  PREMASK bit t = the t-th table IN THAT ORDER is pre-bound.
template <int KIND, int PREMASK>
__device__ __forceinline__ void sc_pf_body(const ScFusedArgs& a, size_t q, const Fr& r, Fr& e0, Fr& e2, Fr& e3, uint32_t bidx, uint32_t nblk) {
  constexpr int NT = KIND == KIND_QUAD ? 2 : KIND == KIND_CUBIC ? 3 : 4;
  constexpr int T0 = KIND == KIND_R1CS ? 1 : 0, T1 = KIND == KIND_R1CS ? 2 : 1, T2 = KIND == KIND_R1CS ? 3 : 2, T3 = 0;
  constexpr bool P0 = (PREMASK & 1) != 0, P1 = (PREMASK & 2) != 0, P2 = (PREMASK & 4) != 0, P3 = (PREMASK & 8) != 0;
  const uint32_t* s0 = a.src[T0]; const uint32_t* s1 = a.src[T1];
  const uint32_t* s2 = a.src[NT > 2 ? T2 : T0]; const uint32_t* s3 = a.src[NT > 3 ? T3 : T0];
  uint32_t* d0 = a.dst[T0]; uint32_t* d1 = a.dst[T1];
  uint32_t* d2 = a.dst[NT > 2 ? T2 : T0]; uint32_t* d3 = a.dst[NT > 3 ? T3 : T0];
  const size_t stride = (size_t)nblk * blockDim.x;
  size_t i = (size_t)bidx * blockDim.x + threadIdx.x;
  if (i >= q) return;
  uint32_t c0 = 0, c2 = 0, c3 = 0;
  ScQuad zn = sc_quad_load<P0>(s0, i, q);
  // the scheduler otherwise moves every prefetch down to its first use (fewer live registers, and the serial chain back):
  // nothing crosses SC_PIN, so the loads stay issued ahead of the arithmetic of the stage they overlap
#define SC_PIN() __builtin_amdgcn_sched_barrier(0)
  for (;;) {
    const size_t inext = i + stride;
    const size_t ipre = inext < q ? inext : i;          // the last iteration re-loads its own first table (harmless) instead of branching
    ScProd P;
    {
      const ScQuad z = zn; zn = sc_quad_load<P1>(s1, i, q); SC_PIN();
      const ScPair t = sc_quad_bind<P0>(z, d0, i, q, r);
      sc_first<KIND>(P, t);
    }
    {
      const ScQuad z = zn;
      if (NT > 2) zn = sc_quad_load<P2>(s2, i, q); else zn = sc_quad_load<P0>(s0, ipre, q);
      SC_PIN();
      const ScPair t = sc_quad_bind<P1>(z, d1, i, q, r);
      sc_times<KIND>(P, t);
    }
    if (NT > 2) {
      const ScQuad z = zn;
      if (NT > 3) zn = sc_quad_load<P3>(s3, i, q); else zn = sc_quad_load<P0>(s0, ipre, q);
      SC_PIN();
      const ScPair t = sc_quad_bind<P2>(z, d2, i, q, r);
      if (KIND == KIND_CUBIC) sc_times<KIND>(P, t); else sc_minus(P, t);
    }
    if (NT > 3) {
      const ScQuad z = zn;
      zn = sc_quad_load<P0>(s0, ipre, q); SC_PIN();
      const ScPair t = sc_quad_bind<P3>(z, d3, i, q, r);
      sc_times<KIND>(P, t);
    }
    fr_acc(e0, P.p0, c0); fr_acc(e2, P.p2, c2);
    if (KIND != KIND_QUAD) fr_acc(e3, P.p3, c3);
    if (inext >= q) break;
    i = inext;
  }
#undef SC_PIN
}
// host side: the masks this kernel is built for (anything else takes the plain kernel)
__host__ __device__ inline bool sc_pf_mask_supported(int kind, unsigned mask) { return mask == 0 || (kind == KIND_CUBIC && mask == 4u); }
template <int KIND>
__global__ void __launch_bounds__(256, 2) k_sc_bind_eval_pf(const ScFusedArgs* __restrict__ args, ScFusedPack pack, size_t q /* old_len / 4 */, ScScalar rmont /* Montgomery form */, uint32_t* __restrict__ partial,
                                                            uint32_t* __restrict__ tickets /* null: the caller runs k_sc_finish */, uint32_t* __restrict__ out, uint32_t seq) {
  ScFusedArgs a;
  if (args) a = args[blockIdx.y];
  else {
#pragma unroll
    for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  }
  const Fr r = fr_from_words(rmont);
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  if (KIND == KIND_CUBIC && a.pre[2]) sc_pf_body<KIND, (KIND == KIND_CUBIC ? 4 : 0)>(a, q, r, e0, e2, e3, blockIdx.x, gridDim.x);   // "par" instance: the shared C arrives bound
  else sc_pf_body<KIND, 0>(a, q, r, e0, e2, e3, blockIdx.x, gridDim.x);
  sc_block_sums_store(e0, e2, e3, KIND != KIND_QUAD, partial, nullptr, 0u);
  if (tickets) sc_last_block_fold(partial, tickets, out, KIND == KIND_QUAD ? 2 : 3, seq);     // (only the flag's system-scope release writes back: once per instance)
}

// out-of-place bind of the top variable of ONE table: dst[i] = Z[i] + r (Z[i + half] - Z[i]), i < half.  Runs ahead of a fused
// round for a table that several instances of the round share (see ScFusedArgs::pre).
__global__ void __launch_bounds__(256) k_bind_oop(const uint32_t* __restrict__ Z, uint32_t* __restrict__ dst, size_t half, ScScalar rmont) {
  const Fr r = fr_from_words(rmont);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    const Fr lo = fe_load<FrP>(Z + 8 * i), hi = fe_load<FrP>(Z + 8 * (i + half));
    fe_store_packed<FrP>(dst + 8 * i, sc_bind1(lo, hi, r));
  }
}

// out[inst][3] canonical; one 64-lane block per instance folds `nblk` partial triples
__global__ void __launch_bounds__(64) k_sc_finish(const uint32_t* __restrict__ partial, int nblk, uint32_t* __restrict__ out) {
  const int inst = blockIdx.x, lane = threadIdx.x;
  for (int q = 0; q < 3; q++) {
    Fr s = fe_zero<FrP>();
    for (int b = lane; b < nblk; b += 64) { Fr x = fe_load<FrP>(partial + 8 * (((size_t)inst * nblk + b) * 3 + q)); s = fe_add(s, x); }     // <= 16 terms below 2.5 r
    s = wave_sum_fr(fe_reduce(s));
    if (lane == 0) fe_store_packed<FrP>(out + 8 * ((size_t)inst * 3 + q), fe_from_mont(s));
  }
}

// bind the top variable of up to `count` tables (blockIdx.y = table) to r (Montgomery form, in `rm`)
__global__ void __launch_bounds__(256) k_bind_top(uint32_t* const* __restrict__ tabs, size_t half, const uint32_t* __restrict__ rm) {
  uint32_t* Z = tabs[blockIdx.y];
  const Fr r = fe_load<FrP>(rm);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Fr lo = fe_gload<FrP>(Z + 8 * i), hi = fe_gload<FrP>(Z + 8 * (i + half));
    fe_gstore_packed<FrP>(Z + 8 * i, sc_bind1(lo, hi, r));
  }
}

// the same with the table pointers and the challenge as kernel arguments (up to BIND_PACK_MAX tables): nothing to stage, nothing to wait for
constexpr int BIND_PACK_MAX = 64;
struct BindPack { uint32_t* t[BIND_PACK_MAX]; };
__global__ void __launch_bounds__(256) k_bind_top_packed(BindPack pack, size_t half, ScScalar rc) {
  uint32_t* Z = nullptr;
#pragma unroll
  for (int i = 0; i < BIND_PACK_MAX; i++) if (i == (int)blockIdx.y) Z = pack.t[i];
  const Fr r = fe_to_mont(fr_from_words(rc));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    Fr lo = fe_gload<FrP>(Z + 8 * i), hi = fe_gload<FrP>(Z + 8 * (i + half));
    fe_gstore_packed<FrP>(Z + 8 * i, sc_bind1(lo, hi, r));
  }
}
// (also run in place: no __restrict__)
__global__ void __launch_bounds__(256) k_fr_to_mont(const uint32_t* in, uint32_t* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fe_store_tab<FrP>(out + 8 * i, fe_to_mont(fe_load<FrP>(in + 8 * i)));
}
// ark-ff's in-memory Fr limbs (Montgomery, R = 2^256) -> this library's table representation (R = 2^261); also run in place
__global__ void __launch_bounds__(256) k_fr_from_ark(const uint32_t* in, uint32_t* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fe_store_tab<FrP>(out + 8 * i, fe_from_ark_mont(fe_load<FrP>(in + 8 * i)));
}
__global__ void __launch_bounds__(256) k_fr_from_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fe_store_packed<FrP>(out + 8 * i, fe_from_mont(fe_load<FrP>(in + 8 * i)));        // canonical plain integers
}

// one level of EqPolynomial::evals: out[2k+1] = in[k]*r_j ; out[2k] = in[k] - out[2k+1]   (hyrax.rs:360-366); r_j in Montgomery form, a kernel argument
__global__ void __launch_bounds__(256) k_eq_level(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t size_in, ScScalar rj) {
  const Fr r = fr_from_words(rj);
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < size_in; k += (size_t)gridDim.x * blockDim.x) {
    Fr s = fe_load<FrP>(in + 8 * k);
    Fr hi = fe_mul(s, r);
    fe_store_tab<FrP>(out + 8 * (2 * k + 1), hi);
    fe_store_tab<FrP>(out + 8 * (2 * k), fe_sub_lazy(s, hi));
  }
}
// two levels in one pass (r_j then r_{j+1}): in[k] -> out[4k .. 4k+3]; saves the intermediate table's write and re-read
__global__ void __launch_bounds__(256) k_eq_level2(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t size_in, ScScalar rj0, ScScalar rj1) {
  const Fr r0 = fr_from_words(rj0), r1 = fr_from_words(rj1);
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < size_in; k += (size_t)gridDim.x * blockDim.x) {
    const Fr s = fe_load<FrP>(in + 8 * k);
    const Fr h = fe_mul(s, r0), l = fe_sub(s, h);
    const Fr hh = fe_mul(h, r1), lh = fe_mul(l, r1);
    fe_store_tab<FrP>(out + 8 * (4 * k), fe_sub_lazy(l, lh));
    fe_store_tab<FrP>(out + 8 * (4 * k + 1), lh);
    fe_store_tab<FrP>(out + 8 * (4 * k + 2), fe_sub_lazy(h, hh));
    fe_store_tab<FrP>(out + 8 * (4 * k + 3), hh);
  }
}
// the first m <= EQ_DIRECT_MAX levels at once: out[i] = prod_j (bit_{m-1-j}(i) ? r_j : 1 - r_j), i < 2^m — the same field element the
// level-by-level recurrence s -> (s - s r, s r) of hyrax.rs:360-366 produces (variable 0 is the most significant index bit).  The point
// travels as a kernel argument, already in Montgomery form (one host product per coordinate): no staging copy, no conversion launch, no
// host wait.  (Statically indexed selects: a dynamic index into a by-value argument would make the compiler copy it to scratch.)
constexpr int EQ_MAX_VARS = 40;
constexpr int EQ_DIRECT_MAX = 12;
struct EqPoint { uint32_t v[EQ_DIRECT_MAX][8]; };
__global__ void __launch_bounds__(256) k_eq_direct(EqPoint rp, int m, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((size_t)1 << m)) return;
  const Fr one = fe_one<FrP>();
  Fr acc = one;
#pragma unroll
  for (int j = 0; j < EQ_DIRECT_MAX; j++) {
    if (j < m) {
      const Fr r = fe_unpack<FrP>(rp.v[j]);
      acc = fe_mul(acc, ((i >> (m - 1 - j)) & 1) ? r : fe_sub(one, r));
    }
  }
  fe_store_tab<FrP>(out + 8 * i, acc);
}
// hash layer: out[j] = (ts[j]+ts_add)*g^2 + val[j]*g + addr[j] - tau.  g2rr = g^2 * R (a "doubly Montgomery" value), so one
// Montgomery product with the plain small integer ts gives mont(ts*g^2); rr = R^2 turns the plain addr into mont(addr).
__global__ void __launch_bounds__(256) k_hash_layer(const uint32_t* __restrict__ addr, const uint32_t* __restrict__ val, const uint32_t* __restrict__ ts, uint32_t ts_add,
                                                    ScScalar g_m, ScScalar g2rr_m, ScScalar ntau_m /* g R, g^2 R^2, (r - tau) R: canonical, from the host */, size_t n, uint32_t* __restrict__ out) {
  const Fr g = fr_from_words(g_m), g2rr = fr_from_words(g2rr_m), ntau = fr_from_words(ntau_m);
  const Fr rr = fe_const_r2<FrP>();
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long av = addr ? addr[j] : (unsigned long long)j;
    const unsigned long long tv = (unsigned long long)(ts ? ts[j] : 0u) + ts_add;
    const Fr a = fe_from_u64<FrP>(av), t = fe_from_u64<FrP>(tv);
    Fr acc = fe_mul(g2rr, t);
    acc = fe_add(acc, fe_mul(fe_load<FrP>(val + 8 * j), g));
    acc = fe_add(acc, fe_mul(rr, a));
    fe_store_tab<FrP>(out + 8 * j, fe_add(acc, ntau));           // three products + (r - tau): (-0.3 r, 4.3 r)
  }
}
// two hashed sets over the same (addr, val) in one pass — the read and write sets of one sparse polynomial (ts and ts + 1), or the init and audit
// sets of a memory (no ts and audit_ts): sparse_mlpoly_full.rs:762-790 builds them in separate loops over the same inputs.  val and addr are read
// once, val * g + addr - tau is shared; per output one more product (ts * g^2).
__global__ void __launch_bounds__(256) k_hash_layer_pair(const uint32_t* __restrict__ addr, const uint32_t* __restrict__ val, const uint32_t* __restrict__ ts_a, uint32_t add_a,
                                                         const uint32_t* __restrict__ ts_b, uint32_t add_b, ScScalar g_m, ScScalar g2rr_m, ScScalar ntau_m, size_t n,
                                                         uint32_t* __restrict__ out_a, uint32_t* __restrict__ out_b) {
  const Fr g = fr_from_words(g_m), g2rr = fr_from_words(g2rr_m), ntau = fr_from_words(ntau_m);
  const Fr rr = fe_const_r2<FrP>();
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long av = addr ? addr[j] : (unsigned long long)j;
    const unsigned long long ta = (unsigned long long)(ts_a ? ts_a[j] : 0u) + add_a, tb = (unsigned long long)(ts_b ? ts_b[j] : 0u) + add_b;
    const Fr base = fe_add(fe_add(fe_mul(fe_load<FrP>(val + 8 * j), g), fe_mul(rr, fe_from_u64<FrP>(av))), ntau);        // two products + (r - tau)
    fe_store_tab<FrP>(out_a + 8 * j, fe_add(base, fe_mul(g2rr, fe_from_u64<FrP>(ta))));                                 // (-0.3 r, 4.3 r) as k_hash_layer
    fe_store_tab<FrP>(out_b + 8 * j, fe_add(base, fe_mul(g2rr, fe_from_u64<FrP>(tb))));
  }
}
__global__ void __launch_bounds__(256) k_product_layer(const uint32_t* __restrict__ in, size_t half, uint32_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
    fe_store_tab<FrP>(out + 8 * i, fe_mul(fe_load<FrP>(in + 8 * i), fe_load<FrP>(in + 8 * (i + half))));
}
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
    for (size_t i = threadIdx.x; i < half; i += blockDim.x) fe_store_tab<FrP>(dst + 8 * i, fe_mul(fe_load<FrP>(src + 8 * i), fe_load<FrP>(src + 8 * (i + half))));
    __threadfence_block();
    __syncthreads();
    src = dst;
    if (half == 1) break;
  }
}
// The same two kernels for up to PC_MANY_MAX circuits of one size at once (blockIdx.y = circuit): the 16 product circuits of a prove (12 over the
// operations' hashed sets, 4 over the memories') are built at the same point (sparse_mlpoly_full.rs:813-823), and below ~2^15 entries a layer's launch
// costs more than its arithmetic — one launch per layer for all circuits instead of one per layer and circuit.
constexpr int PC_MANY_MAX = 16;
struct ProductLayerPack { const uint32_t* in[PC_MANY_MAX]; uint32_t* out[PC_MANY_MAX]; };
__global__ void __launch_bounds__(256) k_product_layer_many(ProductLayerPack pk, size_t half) {
  const uint32_t* in = nullptr; uint32_t* out = nullptr;
#pragma unroll
  for (int i = 0; i < PC_MANY_MAX; i++) if (i == (int)blockIdx.y) { in = pk.in[i]; out = pk.out[i]; }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x)
    fe_store_tab<FrP>(out + 8 * i, fe_mul(fe_load<FrP>(in + 8 * i), fe_load<FrP>(in + 8 * (i + half))));
}
// tails: circuit blockIdx.x, layers out[circuit][k] (uniform selects over the kernel-argument block: a dynamic index would move it to scratch)
struct ProductTailPack { const uint32_t* in[PC_MANY_MAX]; uint32_t* out[PC_MANY_MAX][PT_TAIL_MAX]; };
__global__ void __launch_bounds__(1024) k_product_tail_many(ProductTailPack pk, size_t len) {
  const uint32_t* src = nullptr;
#pragma unroll
  for (int i = 0; i < PC_MANY_MAX; i++) if (i == (int)blockIdx.x) src = pk.in[i];
  int k = 0;
  for (size_t half = len / 2; half >= 1; half >>= 1, k++) {
    uint32_t* dst = nullptr;
#pragma unroll
    for (int i = 0; i < PC_MANY_MAX; i++) {
#pragma unroll
      for (int j = 0; j < PT_TAIL_MAX; j++) if (i == (int)blockIdx.x && j == k) dst = pk.out[i][j];
    }
    for (size_t i = threadIdx.x; i < half; i += blockDim.x) fe_store_tab<FrP>(dst + 8 * i, fe_mul(fe_load<FrP>(src + 8 * i), fe_load<FrP>(src + 8 * (i + half))));
    __threadfence_block();
    __syncthreads();
    src = dst;
    if (half == 1) break;
  }
}
// one running sum per thread -> the block's sum in the 3-slot layout of k_sc_finish (slots 1, 2 zero)
__device__ __forceinline__ void sc_block_sum1_store(Fr acc, uint32_t* __restrict__ o) {
  __shared__ uint32_t sm[4][NL];
  acc = wave_sum_fr(fe_reduce(acc));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) for (int k = 0; k < NL; k++) sm[wv][k] = acc.v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    Fr s = fe_zero<FrP>();
    for (int w = 0; w < 4; w++) { Fr x; for (int k = 0; k < NL; k++) x.v[k] = sm[w][k]; s = fe_add(s, x); }
    fe_store_tab<FrP>(o, fe_reduce(s));
    for (int k = 0; k < 8; k++) { o[8 + k] = 0; o[16 + k] = 0; }
  }
}
// <a, b> partial sums: partial[block] (Montgomery); finished by k_sc_finish-style fold on one value
__global__ void __launch_bounds__(256) k_dot(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, uint32_t* __restrict__ partial) {
  Fr acc = fe_zero<FrP>();
  uint32_t cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fr_acc(acc, fe_mul(fe_load<FrP>(a + 8 * i), fe_load<FrP>(b + 8 * i)), cnt);
  sc_block_sum1_store(acc, partial + 8 * ((size_t)blockIdx.x * 3));
}
// <Z_i, chi> for up to SC_PACK_MAX tables at once (blockIdx.y = table): the evaluations of many polynomials at one point
__global__ void __launch_bounds__(256) k_dot_many(ScArgsPack pack, const uint32_t* __restrict__ chi, size_t n, uint32_t* __restrict__ partial) {
  ScArgs a;
#pragma unroll
  for (int i = 0; i < SC_PACK_MAX; i++) if (i == (int)blockIdx.y) a = pack.a[i];
  const uint32_t* z = a.t[0];
  Fr acc = fe_zero<FrP>();
  uint32_t cnt = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    fr_acc(acc, fe_mul(fe_gload<FrP>(z + 8 * i), fe_load<FrP>(chi + 8 * i)), cnt);
  sc_block_sum1_store(acc, partial + 8 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3));
}
// L*Z: block (column tile of 64, row slice): lanes own columns (coalesced 2 KiB rows), the block's 4 waves split the row
// slice; partial[slice][col] then folded by k_bound_fold.  Z is L_size x R_size row-major.
__global__ void __launch_bounds__(256) k_bound_partial(const uint32_t* __restrict__ Z, const uint32_t* __restrict__ Lv, size_t L_size, size_t R_size, size_t rows_per_slice, uint32_t* __restrict__ partial) {
  const size_t col = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int wv = threadIdx.x >> 6;
  const size_t j0 = (size_t)blockIdx.y * rows_per_slice, j1 = (j0 + rows_per_slice < L_size) ? j0 + rows_per_slice : L_size;
  Fr acc = fe_zero<FrP>();
  uint32_t cnt = 0;
  if (col < R_size) {
    // both operands are table values (non-negative, below 2.5 r): the unsigned product; the next row's element is in flight during the
    // current product (global loads: a flat load would be drained by the next wait, fp.cuh)
    size_t j = j0 + wv;
    Fr zn = fe_zero<FrP>(), ln = zn;
    if (j < j1) { zn = fe_gload<FrP>(Z + 8 * (j * R_size + col)); ln = fe_gload<FrP>(Lv + 8 * j); }
    for (; j < j1; j += 4) {
      const Fr z = zn, l = ln;
      if (j + 4 < j1) { zn = fe_gload<FrP>(Z + 8 * ((j + 4) * R_size + col)); ln = fe_gload<FrP>(Lv + 8 * (j + 4)); }
      fr_acc(acc, fe_mulu(l, z), cnt);
    }
  }
  acc = fe_reduce(acc);
  __shared__ uint32_t sm[4][64][NL];
  for (int k = 0; k < NL; k++) sm[wv][threadIdx.x & 63][k] = acc.v[k];
  __syncthreads();
  if (wv == 0 && col < R_size) {
    Fr s = acc;
    for (int w = 1; w < 4; w++) { Fr x; for (int k = 0; k < NL; k++) x.v[k] = sm[w][threadIdx.x & 63][k]; s = fe_add(s, x); }
    fe_store_tab<FrP>(partial + 8 * ((size_t)blockIdx.y * R_size + col), fe_reduce(s));
  }
}
__global__ void __launch_bounds__(256) k_bound_fold(const uint32_t* __restrict__ partial, size_t nslices, size_t R_size, uint32_t* __restrict__ out) {
  const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= R_size) return;
  Fr s = fe_zero<FrP>();
  uint32_t cnt = 0;
  for (size_t k = 0; k < nslices; k++) fr_acc(s, fe_load<FrP>(partial + 8 * (k * R_size + col)), cnt);
  fe_store_tab<FrP>(out + 8 * col, fe_reduce(s));
}

// out[k*n + i] = mem[k][addr[k][i]]  (AddrTimestamps::deref_mem, sparse_mlpoly_full.rs:245-252), zero padding past count*n.
// With rshift >= 0 only a share of the L x R view (R = 2^rshift) is produced: local row j = row (row0 + j * rstep) of the matrix — the rows
// one device commits when the derefs matrix is dealt over a device group by interleaved rows (out then holds `padded` = nrows * R entries).
struct GatherArgs { const uint32_t* mem; const uint32_t* addr; size_t mem_len; };
__global__ void __launch_bounds__(256) k_gather_merge(const GatherArgs* __restrict__ args, size_t count, size_t n, size_t padded, uint32_t* __restrict__ out, uint32_t* __restrict__ oob,
                                                      int rshift, size_t row0, size_t rstep) {
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < padded; t += (size_t)gridDim.x * blockDim.x) {
    const size_t gt = rshift < 0 ? t : ((((t >> rshift) * rstep + row0) << rshift) | (t & (((size_t)1 << rshift) - 1)));     // index into the whole polynomial
    Fr v = fe_zero<FrP>();
    if (gt < count * n) {
      const size_t k = gt / n, i = gt - k * n;
      const GatherArgs a = args[k];
      const uint32_t idx = a.addr[i];
      if (idx < a.mem_len) v = fe_load<FrP>(a.mem + 8 * (size_t)idx); else atomicAdd(oob, 1u);   // sparse_mlpoly_full.rs:228 assert!(addr < num_cells)
    }
    fe_store_packed<FrP>(out + 8 * t, v);          // a copy of a table value (or zero): stored as it is
  }
}
__global__ void k_fr_set_one(uint32_t* out) { if (threadIdx.x == 0 && blockIdx.x == 0) fe_store_packed<FrP>(out, fe_one<FrP>()); }

// the final claims (sumcheck.rs:302-318): fin[t] (the single entry the last bind left of table t, Montgomery form) as canonical integers
// straight into the host mailbox, flag behind them — no copy, no stream synchronisation (the protocol of the round kernels)
__global__ void __launch_bounds__(128) k_sc_finals(const uint32_t* __restrict__ fin, size_t count, uint32_t* __restrict__ host_out, uint32_t* __restrict__ flag, uint32_t seq) {
  const size_t t = threadIdx.x;
  if (t < count) fe_store_packed<FrP>(host_out + 8 * t, fe_from_mont(fe_load<FrP>(fin + 8 * t)));
  sc_drain_stores();
  __syncthreads();
  if (t == 0) sc_flag_store(flag, seq);
}

// raw words (the XYZZ sums of a few row commitments) and one extra word (the input-check counter) to the host mailbox, flag behind them:
// a 2-row commitment of a bullet round then costs one 4 us launch and a poll instead of two copies and a stream synchronisation
__global__ void __launch_bounds__(256) k_sc_finals_raw(const uint32_t* __restrict__ src, uint32_t nwords, const uint32_t* __restrict__ extra, uint32_t* __restrict__ host_out, uint32_t* __restrict__ flag, uint32_t seq) {
  for (uint32_t t = threadIdx.x; t < nwords; t += 256) host_out[t] = src[t];
  if (threadIdx.x == 0 && extra) host_out[nwords] = *extra;
  sc_drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) sc_flag_store(flag, seq);
}

}  // namespace sbn
