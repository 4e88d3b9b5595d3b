// msm_kernels.cuh — Pippenger bucket MSM over BN254 G1 for gfx950.
//
// Replaces what `G1Projective::msm` does for the reference (src/group.rs:156,173, reached from
// commitments.rs:144-154 and hyrax.rs:253-267).  The algorithm is the bucket method but laid out
// for a 256-CU wave64 machine instead of one CPU thread per window:
//
//   1. k_digits_hist   one thread per scalar: signed radix-2^c digits, histogram of bucket sizes
//   2. k_scan          exclusive scan of the histogram per window (bucket start offsets)
//   3. k_scatter       counting sort: point indices grouped by (window, bucket), sign in bit 31
//   4. k_bucket_acc    one LANE per (window, bucket): XYZZ accumulator in VGPRs, affine points gathered
//                      from HBM (64 B each), no atomics on points, no inter-lane traffic
//   5. k_reduce_l1 / k_reduce_combine   sum_b (b+1)*B_b per window: lane-sequential running sums over L buckets, then a
//                      wave-level suffix scan + tree (DPP/LDS-free __shfl) — log-depth instead of the
//                      reference's 2^c-long serial chain
//   6. host            sum_w 2^(cw) S_w: a 254-doubling serial chain, done on one host core (host_field.hpp)
//
// EC addition is commutative and the final point is returned in canonical affine form, so the order in
// which a bucket's points are added (atomics decide it in step 3) cannot change the result bits.
#pragma once
#include "g1.cuh"

namespace sbn {

struct MsmShape {
  int c;        // window bits
  int W;        // number of windows
  int nb;       // buckets per window = 2^(c-1)
};

// signed digit of window w; carry in/out.  digit in (-2^(c-1), 2^(c-1)]
__device__ __forceinline__ int window_digit(const uint32_t* __restrict__ k /* 8 limbs, global */, int w, int c, uint32_t& carry) {
  const int bit = w * c, limb = bit >> 5, sh = bit & 31;
  uint64_t x = 0;
  if (limb < 8) x = k[limb];
  if (limb + 1 < 8) x |= (uint64_t)k[limb + 1] << 32;
  const uint32_t raw = (uint32_t)(x >> sh) & ((1u << c) - 1u);
  const uint32_t d = raw + carry;
  if (d > (1u << (c - 1))) { carry = 1; return (int)d - (1 << c); }
  carry = 0;
  return (int)d;
}

// Montgomery-form scalars -> canonical integers (only when the caller passes ark-ff's in-memory limbs)
__global__ void __launch_bounds__(256) k_scalars_from_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr a = fe_load<FrP>(in + 8 * i);
  fe_store<FrP>(out + 8 * i, fe_from_mont(a));
}
// canonical affine points -> Montgomery coordinates
__global__ void __launch_bounds__(256) k_points_to_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fq x = fe_load<FqP>(in + 16 * i), y = fe_load<FqP>(in + 16 * i + 8);
  fe_store<FqP>(out + 16 * i, fe_to_mont(x));
  fe_store<FqP>(out + 16 * i + 8, fe_to_mont(y));
}

// 1. histogram.  `stride` = distance in scalars between rows' scalars (single MSM: rows = 1).
__global__ void __launch_bounds__(256) k_digits_hist(const uint32_t* __restrict__ scalars, size_t n, MsmShape s, uint32_t* __restrict__ hist) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* k = scalars + 8 * i;
  uint32_t carry = 0;
  for (int w = 0; w < s.W; w++) {
    int d = window_digit(k, w, s.c, carry);
    if (d != 0) atomicAdd(&hist[(size_t)w * s.nb + (d < 0 ? -d : d) - 1], 1u);
  }
}

// 2. exclusive scan per window: offs[w][b] = sum_{b'<b} hist[w][b'];  cursor = copy of offs.  One block per window.
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor, int nb) {
  __shared__ uint32_t part[1024];
  const int w = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int per = (nb + T - 1) / T;
  const uint32_t* h = hist + (size_t)w * nb;
  uint32_t sum = 0;
  for (int j = 0; j < per; j++) { int b = t * per + j; if (b < nb) sum += h[b]; }
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {   // Hillis-Steele inclusive scan of the per-thread sums
    uint32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - sum;
  for (int j = 0; j < per; j++) {
    int b = t * per + j;
    if (b < nb) { offs[(size_t)w * nb + b] = run; cursor[(size_t)w * nb + b] = run; run += h[b]; }
  }
}

// 3. counting-sort scatter
__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ scalars, size_t n, MsmShape s, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* k = scalars + 8 * i;
  uint32_t carry = 0;
  for (int w = 0; w < s.W; w++) {
    int d = window_digit(k, w, s.c, carry);
    if (d != 0) {
      uint32_t pos = atomicAdd(&cursor[(size_t)w * s.nb + (d < 0 ? -d : d) - 1], 1u);
      sorted[(size_t)w * n + pos] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
    }
  }
}

// 4. bucket accumulation: lane per (window, bucket)
__global__ void __launch_bounds__(256) k_bucket_acc(const uint32_t* __restrict__ bases /* n x 16 u32, Montgomery affine */, size_t n, MsmShape s,
                                                    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs,
                                                    const uint32_t* __restrict__ sorted, uint32_t* __restrict__ buckets /* W*nb x 32 u32 */) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)s.W * s.nb) return;
  const int w = (int)(t / s.nb);
  const uint32_t cnt = hist[t], start = offs[t];
  const uint32_t* lst = sorted + (size_t)w * n + start;
  XYZZ acc = xyzz_inf();
  for (uint32_t j = 0; j < cnt; j++) {
    const uint32_t e = lst[j];
    const Affine p = aff_load(bases + 16 * (size_t)(e & 0x7fffffffu));
    xyzz_madd(acc, p, (e >> 31) != 0);
  }
  xyzz_store(buckets + 32 * t, acc);
}

// ---- weighted bucket sums --------------------------------------------------------------------
// wave-level helpers: every lane holds one point
__device__ __forceinline__ XYZZ wave_suffix_scan(XYZZ v, int lane) {   // v_l <- sum_{l' >= l} v_l'
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    XYZZ o = xyzz_shfl_down(v, d);
    if (lane + d < 64) v = xyzz_add(v, o);
  }
  return v;
}
__device__ __forceinline__ XYZZ wave_sum(XYZZ v, int lane) {           // lane 0 <- sum of all lanes
#pragma unroll 1
  for (int d = 32; d >= 1; d >>= 1) {
    XYZZ o = xyzz_shfl_down(v, d);
    if (lane < d) v = xyzz_add(v, o);
  }
  return v;
}
__device__ __forceinline__ XYZZ xyzz_mul_pow2(XYZZ v, int k) {
#pragma unroll 1
  for (int i = 0; i < k; i++) v = xyzz_dbl(v);
  return v;
}

// level 1: one wave per chunk of 64*L consecutive buckets of one problem (window).
//   S  = sum_i X_i ,  Wt = sum_i i * X_i   (i = 0-based index inside the chunk)
// out[chunk] = {S, Wt}
__global__ void __launch_bounds__(64) k_reduce_l1(const uint32_t* __restrict__ X, int L, int logL, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x;
  const size_t chunk = blockIdx.x;
  const uint32_t* base = X + 32 * ((chunk * 64 + lane) * (size_t)L);
  // lane-sequential running sums over its L buckets: run = sum X_i, acc = sum i*X_i (local i)
  XYZZ run = xyzz_inf(), acc = xyzz_inf();
  for (int i = L - 1; i >= 1; i--) {
    run = xyzz_add(run, xyzz_load(base + 32 * (size_t)i));
    acc = xyzz_add(acc, run);
  }
  run = xyzz_add(run, xyzz_load(base));
  // across lanes: index = lane*L + local  =>  Wt = sum_l acc_l + L * sum_l l * run_l
  XYZZ suf = wave_suffix_scan(run, lane);          // suf_l = sum_{l'>=l} run_l'
  XYZZ S = suf;                                    // lane 0 holds the chunk total
  XYZZ term = (lane >= 1) ? suf : xyzz_inf();      // sum_{l>=1} suf_l = sum_l l*run_l
  term = xyzz_mul_pow2(term, logL);
  term = xyzz_add(term, acc);
  XYZZ Wt = wave_sum(term, lane);
  if (lane == 0) { xyzz_store(out + 64 * chunk, S); xyzz_store(out + 64 * chunk + 32, Wt); }
}

// combine: one wave per group of 64 consecutive chunk results of one problem; lane l holds chunk g0+l, each chunk
// standing for M = 2^logM buckets.  Produces the {S, Wt} of the 64*M-bucket super-chunk:
//   S' = sum_l S_l ,  Wt' = sum_l Wt_l + M * sum_l l*S_l
// G = chunks per problem at the input level (lanes past G contribute the identity); Gout = ceil(G/64).
// When `final` is set (Gout == 1) the wave writes sum_b (b+1) X_b = Wt' + S' as a single point instead.
__global__ void __launch_bounds__(64) k_reduce_combine(const uint32_t* __restrict__ in, int G, int Gout, int logM, int final, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x;
  const size_t prob = blockIdx.x / Gout;
  const int grp = blockIdx.x % Gout;
  const int g = grp * 64 + lane;
  XYZZ S = xyzz_inf(), Wt = xyzz_inf();
  if (g < G) {
    const uint32_t* p = in + 64 * (prob * (size_t)G + g);
    S = xyzz_load(p); Wt = xyzz_load(p + 32);
  }
  XYZZ suf = wave_suffix_scan(S, lane);
  XYZZ term = (lane >= 1) ? suf : xyzz_inf();
  term = xyzz_mul_pow2(term, logM);
  term = xyzz_add(term, Wt);
  XYZZ W2 = wave_sum(term, lane);
  if (lane == 0) {
    if (final) {
      xyzz_store(out + 32 * prob, xyzz_add(W2, suf));
    } else {
      uint32_t* o = out + 64 * (prob * (size_t)Gout + grp);
      xyzz_store(o, suf); xyzz_store(o + 32, W2);
    }
  }
}

// XYZZ -> affine for `count` points, one lane each (Fermat inversion): Montgomery coordinates for resident tables
// and/or canonical little-endian bytes for the caller; either output may be null.
__global__ void __launch_bounds__(64) k_xyzz_to_affine(const uint32_t* __restrict__ in, uint32_t* __restrict__ out_mont, uint32_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  XYZZ p = xyzz_load(in + 32 * i);
  Affine a = xyzz_to_affine(p);            // (0,0) for infinity
  if (out_mont) aff_store(out_mont + 16 * i, a);
  if (out_xy) { fe_store<FqP>(out_xy + 16 * i, fe_from_mont(a.x)); fe_store<FqP>(out_xy + 16 * i + 8, fe_from_mont(a.y)); }
  if (out_inf) out_inf[i] = xyzz_is_inf(p) ? 1 : 0;
}

// fixed-base multiples s_i * G of the generator (1,2): one lane per scalar, double-and-add MSB first.
// Setup only (MultiCommitGens::new, commitments.rs:31-62 -> group.rs:110-131); ~66 % of the reference's scalars are 1.
__global__ void __launch_bounds__(64) k_mul_generator(const uint32_t* __restrict__ scalars /* canonical */, size_t n, uint32_t* __restrict__ out_xyzz) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* k = scalars + 8 * i;
  Affine G; G.x = fe_one<FqP>(); G.y = fe_dbl(G.x);
  XYZZ acc = xyzz_inf();
  int top = 255;
  while (top >= 0 && !((k[top >> 5] >> (top & 31)) & 1u)) top--;
  for (int b = top; b >= 0; b--) {
    acc = xyzz_dbl(acc);
    if ((k[b >> 5] >> (b & 31)) & 1u) xyzz_madd(acc, G, false);
  }
  xyzz_store(out_xyzz + 32 * i, acc);
}

}  // namespace sbn
