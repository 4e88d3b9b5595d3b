// comb_kernels.cuh — fixed-base direct lookup for the Hyrax row commits (hyrax.rs:253-267 over a fixed MultiCommitGens).
//
// The generators of a commitment key never change, and an MI355X has 288 GB of HBM.  So beyond the window table
// 2^(c w) * G_j the key can hold EVERY digit multiple:  T[w][j][d-1] = d * 2^(c w) * G_j,  d = 1 .. 2^(c-1)  (affine,
// Montgomery, 64 B; negative digits negate y).  A row commitment is then sum_j sum_w (+-)T[w][j][|d_jw|]: W mixed additions
// per scalar into one accumulator — no buckets, no sort, no bucket reduction, and therefore no reason to keep c small:
// at c = 16 a 254-bit scalar costs 16 mixed additions instead of the 24 the bucket method's optimum (c = 11) needs for the
// 2813 unique reference generators.  The table for that set is 2814 x 16 x 32768 x 64 B = 94 GB; random 64-byte gathers
// from a table of that size run at 1.9e10 /s on MI355X (tools/micro/gather.hip), above the 1.2e10 /s mixed-addition ceiling.
#pragma once
#include "msm_kernels.cuh"

namespace sbn {

constexpr int COMB_CH = 64;      // multiples per lane in the table build (one batched inversion per lane)
constexpr int COMB_C_MAX = 17;   // widest lookup window (c = 17: 15 windows; 177 GB for the 2814 unique points of the 8193-generator set)

// One window slab of the table.  Lane t -> (column j, chunk a): entries d = 64 a + 1 .. 64 a + 64 of column j.
// tmp_xyzz / tmp_pref: COMB_CH x lanes records (k-major, so a wave's accesses are contiguous).
__global__ void __launch_bounds__(64) k_comb_build(const uint32_t* __restrict__ wtab /* this window's npts affine points */, size_t npts, int c, size_t lane0, size_t lanes,
                                                   uint32_t* __restrict__ tmp_xyzz, uint32_t* __restrict__ tmp_pref, uint32_t* __restrict__ slab /* npts << (c-1) affine */) {
  const size_t tl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tl >= lanes) return;
  const size_t t = lane0 + tl;
  const size_t A = ((size_t)1 << (c - 1)) / COMB_CH;
  const size_t j = t / A, a = t - j * A;
  const Affine P = aff_load(wtab + 16 * j);
  uint32_t* dst = slab + 16 * ((j << (c - 1)) + a * COMB_CH);
  if (aff_is_inf(P)) {                                   // a generator at infinity: all its multiples are
    const Affine z = P;
    for (int k = 0; k < COMB_CH; k++) aff_store(dst + 16 * k, z);
    return;
  }
  // Q = (64 a) * P, then 64 mixed additions
  XYZZ Q = xyzz_inf();
  const uint32_t m0 = (uint32_t)(a * COMB_CH);
  for (int b = COMB_C_MAX - 1; b >= 0; b--) {
    Q = xyzz_dbl(Q);
    if ((m0 >> b) & 1u) xyzz_madd(Q, P, false);
  }
  Fq run = fe_one<FqP>();
  for (int k = 0; k < COMB_CH; k++) {
    xyzz_madd(Q, P, false);                              // (64 a + k + 1) * P: never infinity (prime order, multiplier < r)
    xyzz_store(tmp_xyzz + 32 * ((size_t)k * lanes + tl), Q);
    run = fe_mul(run, fe_mul(Q.ZZ, Q.ZZZ));
    fe_store_packed<FqP>(tmp_pref + 8 * ((size_t)k * lanes + tl), fe_fix_nonneg<FqP, 1>(run));
  }
  Fq inv = fe_inv(run);
  for (int k = COMB_CH - 1; k >= 0; k--) {
    const XYZZ E = xyzz_load(tmp_xyzz + 32 * ((size_t)k * lanes + tl));
    const Fq prev = k ? fe_load<FqP>(tmp_pref + 8 * ((size_t)(k - 1) * lanes + tl)) : fe_one<FqP>();
    const Fq I = fe_mul(inv, prev);                      // 1 / (ZZ_k * ZZZ_k)
    inv = fe_mul(inv, fe_mul(E.ZZ, E.ZZZ));
    Affine o; o.x = fe_mul(E.X, fe_mul(I, E.ZZZ)); o.y = fe_mul(E.Y, fe_mul(I, E.ZZ));
    aff_store(dst + 16 * k, o);
  }
}

// Row commits by lookup.  grid = (rows, S): S blocks share a row (few rows: keep the chip busy, keep chains short); lane tid of
// block (row, s) sums the columns col = s*256 + tid, + 256 S, ... and stores its accumulator to partial[(row*S + s)*256 + tid].
// The kernel is nothing but the accumulate loop (the tree sum lives in k_comb_fold: keeping it out holds this kernel at the
// register budget of the bucket accumulate kernel, 3 waves per SIMD instead of 2).
// Rows flagged constant or zero (flags[row] != 0) are left to k_comb_rows_const: their merged form has at most two non-zero
// scalars, and walking all columns just to find zero digits is a chain of exposed load latencies.
__global__ void __launch_bounds__(256, 3) k_comb_rows(const uint32_t* __restrict__ table, DigitArgs a, MsmShape s, const uint8_t* __restrict__ flags, uint32_t* __restrict__ partial) {
  const size_t row = blockIdx.x; const unsigned S = gridDim.y;
  if (flags && flags[row] != 0) return;
  XYZZ acc = xyzz_inf();
  {
    Affine p_prev; bool neg_prev = false, have = false;
    for (size_t col = (size_t)blockIdx.y * blockDim.x + threadIdx.x; col < a.n; col += (size_t)S * blockDim.x) {
      const uint32_t* k = (col < a.R) ? a.scalars + 8 * (row * a.R + col) : a.blinds + 8 * row;
      { const uint4 q0 = reinterpret_cast<const uint4*>(k)[0], q1 = reinterpret_cast<const uint4*>(k)[1];
        const uint32_t kk[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        if (!fe_is_canonical<FrP>(kk)) atomicAdd(a.bad, 1u); }      // scalar >= r: reported by the host wrapper, never summed silently
      uint32_t carry = 0;
      for (int w = 0; w < s.W; w++) {
        const int d = window_digit(k, w, s.c, carry);
        if (d == 0) continue;
        const size_t idx = ((((size_t)w * a.tstride + col)) << (s.c - 1)) + (size_t)((d < 0 ? -d : d) - 1);
        const Affine p = aff_load(table + 16 * idx);     // in flight during the previous point's mixed addition
        if (have) xyzz_madd(acc, p_prev, neg_prev);
        p_prev = p; neg_prev = d < 0; have = true;
      }
    }
    if (have) xyzz_madd(acc, p_prev, neg_prev);
  }
  xyzz_store(partial + 32 * ((row * S + blockIdx.y) * blockDim.x + threadIdx.x), acc);
}
// Constant / zero rows of a merged matrix (k_merge_small): only column col_a (the sum-of-all-bases column, carrying the row's value)
// and col_b (the column h merged into, carrying the blind; ~0 = none) can be non-zero.  One wave per row, a lane per (column, window).
__global__ void __launch_bounds__(64) k_comb_rows_const(const uint32_t* __restrict__ table, DigitArgs a, MsmShape s, const uint8_t* __restrict__ flags, int skip_zero,
                                                        size_t col_a, size_t col_b, uint32_t* __restrict__ sparse) {
  const size_t row = blockIdx.x; const int lane = threadIdx.x;
  if (flags[row] == 0) return;
  XYZZ acc = xyzz_inf();
  if (!(skip_zero && flags[row] == 2)) {
    const int n2 = (col_b == ~(size_t)0) ? s.W : 2 * s.W;
    for (int idx = lane; idx < n2; idx += 64) {
      const size_t col = idx < s.W ? col_a : col_b; const int w = idx < s.W ? idx : idx - s.W;
      const uint32_t* k = (col < a.R) ? a.scalars + 8 * (row * a.R + col) : a.blinds + 8 * row;
      if (w == 0) { const uint4 q0 = reinterpret_cast<const uint4*>(k)[0], q1 = reinterpret_cast<const uint4*>(k)[1];
        const uint32_t kk[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        if (!fe_is_canonical<FrP>(kk)) atomicAdd(a.bad, 1u); }      // scalar >= r: reported by the host wrapper, never summed silently
      const int d = window_digit_indep(k, w, s.c);
      if (d == 0) continue;
      const size_t ti = ((((size_t)w * a.tstride + col)) << (s.c - 1)) + (size_t)((d < 0 ? -d : d) - 1);
      xyzz_madd(acc, aff_load(table + 16 * ti), d < 0);
    }
  }
  acc = wave_sum_all(acc);
  if (lane == 0) xyzz_store(sparse + 32 * row, acc);
}
// Few rows (bullet rounds, single commits): the (window, column) pairs of a row are dealt out one by one over S*256 lanes, so
// a lane's chain is ncol*W / (256 S) mixed additions however few columns there are, and the block sums its accumulators.
__global__ void __launch_bounds__(256) k_comb_rows_flat(const uint32_t* __restrict__ table, DigitArgs a, MsmShape s, const uint8_t* __restrict__ skip, uint32_t* __restrict__ partial) {
  const size_t row = blockIdx.x; const unsigned S = gridDim.y;
  XYZZ acc = xyzz_inf();
  if (!(skip && skip[row] == 2)) {
    const uint32_t ncol = (uint32_t)a.n, total = ncol * (uint32_t)s.W;
    // one mixed-addition site (the kernel runs every instruction once: its size is instruction-fetch time); the next point is
    // in flight during the previous point's addition
    Affine p_prev; bool neg_prev = false, have = false;
    uint32_t idx = blockIdx.y * blockDim.x + threadIdx.x; bool more = idx < total;
    while (more || have) {
      Affine p; bool neg = false, got = false;
      if (more) {
        const uint32_t w = idx / ncol, col = idx - w * ncol;
        const uint32_t* k = (col < a.R) ? a.scalars + 8 * (row * a.R + col) : a.blinds + 8 * row;
        if (w == 0) { const uint4 q0 = reinterpret_cast<const uint4*>(k)[0], q1 = reinterpret_cast<const uint4*>(k)[1];
          const uint32_t kk[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
          if (!fe_is_canonical<FrP>(kk)) atomicAdd(a.bad, 1u); }      // scalar >= r: reported by the host wrapper, never summed silently
        const int d = window_digit_indep(k, (int)w, s.c);
        if (d != 0) {
          const size_t ti = ((((size_t)w * a.tstride + col)) << (s.c - 1)) + (size_t)((d < 0 ? -d : d) - 1);
          p = aff_load(table + 16 * ti); neg = d < 0; got = true;
        }
        idx += S * blockDim.x; more = idx < total;
      }
      if (have) xyzz_madd(acc, p_prev, neg_prev);
      p_prev = p; neg_prev = neg; have = got;
    }
  }
  __shared__ uint32_t sm[4][32];
  const XYZZ t = block_sum_quad(acc, sm);                 // log-depth, quad-cooperative additions (g1.cuh)
  if (threadIdx.x == 0) xyzz_store(partial + 32 * (row * S + blockIdx.y), t);
}
// per row: `per_row` lane accumulators -> out[row].  One block per row: strided partial sums, wave tree, and — when the block has
// four waves (many accumulators per row, i.e. few rows) — an LDS step across them.  Many rows use one wave per row.
// Rows with flags[row] != 0 take the point k_comb_rows_const left in sparse[row].
__global__ void __launch_bounds__(256) k_comb_fold(const uint32_t* __restrict__ partial, unsigned per_row, uint32_t* __restrict__ out, const uint8_t* __restrict__ flags, const uint32_t* __restrict__ sparse) {
  const size_t row = blockIdx.x;
  __shared__ uint32_t sm[4][32];
  if (flags && flags[row] != 0) { if (threadIdx.x == 0) xyzz_store(out + 32 * row, xyzz_load(sparse + 32 * row)); return; }
  XYZZ v = xyzz_inf();
  // (inlined addition: the out-of-line xyzz_add would give this latency-bound kernel a scratch segment)
  for (unsigned i = threadIdx.x; i < per_row; i += blockDim.x) v = xyzz_add_inl(v, xyzz_load(partial + 32 * (row * per_row + i)));
  v = block_sum_quad(v, sm);
  if (threadIdx.x == 0) xyzz_store(out + 32 * row, v);
}

}  // namespace sbn
