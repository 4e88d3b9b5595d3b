// msm_kernels.cuh — Pippenger bucket MSM over BN254 G1 for gfx950.
//
// Replaces what `G1Projective::msm` does for the reference (src/group.rs:156,173, reached from
// commitments.rs:144-154 and hyrax.rs:253-267).  The algorithm is the bucket method but laid out
// for a 256-CU wave64 machine instead of one CPU thread per window:
//
//   1. k_digits<hist>  one thread per scalar: signed radix-2^c digits, histogram of bucket sizes
//   2. k_scan          exclusive scan of the histogram per problem (bucket start offsets)
//   3. k_digits<scat>  counting sort: point indices grouped by (problem, bucket), sign in bit 31
//   4. k_acc_*         one LANE per bucket: XYZZ accumulator in VGPRs, affine points gathered from HBM (64 B
//                      each), no atomics on points; oversized buckets are cut into segments (k_acc_extra/merge)
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

// signed digit of window w; carry in/out.  digit in [-2^(c-1), 2^(c-1)) (ark-ec's recentring)
__device__ __forceinline__ int window_digit(const uint32_t* __restrict__ k /* 8 limbs, global */, int w, int c, uint32_t& carry) {
  const int bit = w * c, limb = bit >> 5, sh = bit & 31;
  uint64_t x = 0;
  if (limb < 8) x = k[limb];
  if (limb + 1 < 8) x |= (uint64_t)k[limb + 1] << 32;
  const uint32_t raw = (uint32_t)(x >> sh) & ((1u << c) - 1u);
  const uint32_t d = raw + carry;
  if (d >= (1u << (c - 1))) { carry = 1; return (int)d - (1 << c); }
  carry = 0;
  return (int)d;
}

// The same recoding without the carry chain: the carry INTO window w is bit (c*w - 1) of k (the lower windows represent
// k mod 2^(cw) recentred into [-2^(cw-1), 2^(cw-1)), which borrows exactly when that bit is set), so every window can be
// taken on its own.  Digits land in [-2^(c-1), 2^(c-1)] — the sequential rule's -2^(c-1)-with-carry corner comes out as
// +2^(c-1) without one; both are the same number.  Used where the windows of one scalar are spread over lanes.
__device__ __forceinline__ int window_digit_indep(const uint32_t* __restrict__ k, int w, int c) {
  const int bit = w * c, limb = bit >> 5, sh = bit & 31;
  uint64_t x = 0;
  if (limb < 8) x = k[limb];
  if (limb + 1 < 8) x |= (uint64_t)k[limb + 1] << 32;
  const uint32_t raw = (uint32_t)(x >> sh) & ((1u << c) - 1u);
  const uint32_t carry = bit ? (k[(bit - 1) >> 5] >> ((bit - 1) & 31)) & 1u : 0u;
  const int d = (int)(raw + carry);
  return (raw >> (c - 1)) ? d - (1 << c) : d;
}

// Montgomery-form scalars -> canonical integers (only when the caller passes ark-ff's in-memory limbs)
__global__ void __launch_bounds__(256) k_scalars_from_mont(const uint32_t* in, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr a = fe_load<FrP>(in + 8 * i);                            // ark-ff's limbs: value * 2^256
  fe_store_packed<FrP>(out + 8 * i, fe_ark_mont_to_plain(a));
}
// this library's Montgomery tables (R = 2^261, lazy representatives) -> canonical integers (a table fed to a commitment)
__global__ void __launch_bounds__(256) k_scalars_from_internal(const uint32_t* in, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_store_packed<FrP>(out + 8 * i, fe_from_mont(fe_load<FrP>(in + 8 * i)));
}
// canonical affine points -> Montgomery coordinates
// (also run in place: no __restrict__)
__global__ void __launch_bounds__(256) k_points_to_mont(const uint32_t* in, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fq x = fe_load<FqP>(in + 16 * i), y = fe_load<FqP>(in + 16 * i + 8);
  fe_store_packed<FqP>(out + 16 * i, fe_canon_small(fe_to_mont(x)));
  fe_store_packed<FqP>(out + 16 * i + 8, fe_canon_small(fe_to_mont(y)));
}
// ark-ff's in-memory affine coordinates (Montgomery, R = 2^256) -> this library's (R = 2^261); also run in place
__global__ void __launch_bounds__(256) k_points_from_ark(const uint32_t* in, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fq x = fe_load<FqP>(in + 16 * i), y = fe_load<FqP>(in + 16 * i + 8);
  fe_store_packed<FqP>(out + 16 * i, fe_canon_small(fe_from_ark_mont(x)));
  fe_store_packed<FqP>(out + 16 * i + 8, fe_canon_small(fe_from_ark_mont(y)));
}

// ---- digit extraction / counting sort ----------------------------------------------------------
// Two front-ends share every later stage.  A "problem" owns nb buckets and produces one weighted bucket sum:
//   MODE_SINGLE  one MSM of n terms: problem = window w (W problems); entry = point index
//   MODE_ROWS    Hyrax row commits over shared bases (hyrax.rs:253-267): problem = matrix row (L problems); all W
//                windows of a row share ONE bucket set because the table holds 2^(c*w)*G_j for every window w
//                (entry = w*tstride + column), so no doublings are needed after the bucket sum.
enum { MODE_SINGLE = 0, MODE_ROWS = 1 };
struct DigitArgs {
  const uint32_t* scalars;   // SINGLE: n x 8 limbs.  ROWS: L x R x 8, row-major
  const uint32_t* blinds;    // ROWS: L x 8 or null (null = zero blinds, hyrax.rs:301-305)
  size_t n;                  // SINGLE: number of scalars.  ROWS: columns per row incl. the blind column if present
  size_t R;                  // ROWS: row length of Z
  size_t L;                  // ROWS: number of rows
  size_t tstride;            // ROWS: points per window slab of the table
  size_t estride;            // sorted-entry capacity per problem
  uint32_t* bad;             // device counter of scalars >= r met while reading the caller's input (never null)
};

// signed digits of every scalar, stored once: dig[p*E + e].  Window bits never exceed 16 (LDS counter capacity), so a
// digit in [-2^15, 2^15) fits 16 bits: half the bytes of the three passes that stream the digit array.
typedef int16_t dig_t;
constexpr int MSM_C_MAX = 16;
//   SINGLE: p = window w, e = scalar index          ROWS: p = row, e = w*ncol + col
template <int MODE>
__global__ void __launch_bounds__(256) k_digits_store(DigitArgs a, MsmShape s, dig_t* __restrict__ dig, const uint8_t* __restrict__ skip) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t* k; size_t row = 0, col = 0;
  if (MODE == MODE_SINGLE) {
    if (t >= a.n) return;
    k = a.scalars + 8 * t;
  } else {
    if (t >= a.L * a.n) return;
    row = t / a.n; col = t - row * a.n;
    if (skip && skip[row] == 2) return;                  // all-zero row: k_sort_rows never reads its digits
    k = (col < a.R) ? a.scalars + 8 * (row * a.R + col) : a.blinds + 8 * row;
  }
  { uint32_t kk[8]; const uint4 k0 = reinterpret_cast<const uint4*>(k)[0], k1 = reinterpret_cast<const uint4*>(k)[1];
    kk[0] = k0.x; kk[1] = k0.y; kk[2] = k0.z; kk[3] = k0.w; kk[4] = k1.x; kk[5] = k1.y; kk[6] = k1.z; kk[7] = k1.w;
    if (!fe_is_canonical<FrP>(kk)) atomicAdd(a.bad, 1u); }   // a value >= r would drop the top window's carry: reported, never summed silently
  uint32_t carry = 0;
  for (int w = 0; w < s.W; w++) {
    const int d = window_digit(k, w, s.c, carry);
    if (MODE == MODE_SINGLE) dig[(size_t)w * a.n + t] = (dig_t)d;
    else dig[row * a.estride + (size_t)w * a.n + col] = (dig_t)d;
  }
}

// ---- counting sort with LDS-resident counters --------------------------------------------------
// Random global atomics run at ~2e10/s on MI355X (one dword per lane in 64 different rows); the same
// increments on LDS counters are ~100x cheaper.  A block owns (problem p, bucket range r, entry chunk k): it
// histograms its chunk into LDS, the per-block counts are prefix-summed over k, and the scatter pass replays the
// chunk against LDS cursors.  The order inside a bucket depends on LDS arbitration only; sums do not (EC addition
// commutes and results are compared in canonical affine form).
struct SortGeom {
  size_t E;          // digit entries per problem
  size_t estride;    // sorted-entry capacity per problem (== E)
  size_t chunk;      // entries per chunk
  size_t ncol, tstride;   // ROWS payload mapping: e = w*ncol + col -> point index w*tstride + col
  int nb, RS, logRS, R, K, mode;
};
extern __shared__ uint32_t sort_lds[];

__global__ void __launch_bounds__(1024) k_hist_lds(const dig_t* __restrict__ dig, SortGeom g, uint32_t* __restrict__ blockhist) {
  const int k = blockIdx.x, r = blockIdx.y; const size_t p = blockIdx.z;
  for (int j = threadIdx.x; j < g.RS; j += blockDim.x) sort_lds[j] = 0;
  __syncthreads();
  const size_t e0 = (size_t)k * g.chunk, e1 = (e0 + g.chunk < g.E) ? e0 + g.chunk : g.E;
  const dig_t* d = dig + p * g.E;
  for (size_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
    const int v = d[e];
    if (v == 0) continue;
    const int b = (v < 0 ? -v : v) - 1;
    if ((b >> g.logRS) == r) atomicAdd(&sort_lds[b & (g.RS - 1)], 1u);
  }
  __syncthreads();
  uint32_t* o = blockhist + ((p * g.R + r) * g.K + k) * (size_t)g.RS;
  for (int j = threadIdx.x; j < g.RS; j += blockDim.x) o[j] = sort_lds[j];
}
// exclusive prefix over the K chunks of every bucket (in place) and the bucket totals
__global__ void __launch_bounds__(256) k_block_prefix(uint32_t* __restrict__ blockhist, SortGeom g, size_t nbuckets, uint32_t* __restrict__ hist) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbuckets) return;
  const size_t p = t / g.nb; const int b = (int)(t - p * g.nb);
  const int r = b >> g.logRS, j = b & (g.RS - 1);
  uint32_t* q = blockhist + ((p * g.R + r) * g.K) * (size_t)g.RS + j;
  uint32_t run = 0;
  for (int k = 0; k < g.K; k++) { const uint32_t v = q[(size_t)k * g.RS]; q[(size_t)k * g.RS] = run; run += v; }
  hist[t] = run;
}
__global__ void __launch_bounds__(1024) k_scatter_lds(const dig_t* __restrict__ dig, SortGeom g, const uint32_t* __restrict__ blockhist, const uint32_t* __restrict__ offs, uint32_t* __restrict__ sorted) {
  const int k = blockIdx.x, r = blockIdx.y; const size_t p = blockIdx.z;
  const uint32_t* base = blockhist + ((p * g.R + r) * g.K + k) * (size_t)g.RS;
  const uint32_t* off = offs + p * g.nb + (size_t)r * g.RS;
  for (int j = threadIdx.x; j < g.RS; j += blockDim.x) sort_lds[j] = off[j] + base[j];
  __syncthreads();
  const dig_t* d = dig + p * g.E;
  uint32_t* out = sorted + p * g.estride;
  if (g.mode == MODE_SINGLE) {
    const size_t e0 = (size_t)k * g.chunk, e1 = (e0 + g.chunk < g.E) ? e0 + g.chunk : g.E;
    for (size_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
      const int v = d[e];
      if (v == 0) continue;
      const int b = (v < 0 ? -v : v) - 1;
      if ((b >> g.logRS) != r) continue;
      const uint32_t pos = atomicAdd(&sort_lds[b & (g.RS - 1)], 1u);
      out[pos] = (uint32_t)e | (v < 0 ? 0x80000000u : 0u);
    }
  } else {
    // ROWS: entry e = w*ncol + col -> table index w*tstride + col (E < 2^31, so the division is 32-bit)
    const uint32_t ncol = (uint32_t)g.ncol, ts = (uint32_t)g.tstride;
    const uint32_t e0 = (uint32_t)((size_t)k * g.chunk), e1 = (uint32_t)((e0 + g.chunk < g.E) ? e0 + g.chunk : g.E);
    for (uint32_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
      const int v = d[e];
      if (v == 0) continue;
      const int b = (v < 0 ? -v : v) - 1;
      if ((b >> g.logRS) != r) continue;
      const uint32_t pos = atomicAdd(&sort_lds[b & (g.RS - 1)], 1u);
      const uint32_t w = e / ncol;
      out[pos] = (w * ts + (e - w * ncol)) | (v < 0 ? 0x80000000u : 0u);
    }
  }
}

// ROWS fast path: one block sorts one whole row (problem) — histogram, scan and scatter fused, everything but the digits
// and the result in LDS.  The scatter is staged: scattered 4-byte global stores cost one L2 transaction each (measured
// ~5.6e10 entries/s chip-wide, 3.4 ms of the 25.7 ms derefs commit), so sorted positions are cut into ranges of SL
// entries (by the start offset of their bucket), each range is assembled in LDS and flushed with coalesced stores.
constexpr uint32_t SORT_SL_LOG = 15, SORT_SL = 1u << SORT_SL_LOG, SORT_SLACK = 1024, SORT_MAXR = 64;
__host__ __device__ inline size_t sort_rows_lds_bytes(int nb) { return ((size_t)nb + (size_t)nb / 4 + 1024 + SORT_MAXR + 2 + SORT_SL + SORT_SLACK) * 4; }
__global__ void __launch_bounds__(1024) k_sort_rows(const dig_t* __restrict__ dig, SortGeom g, uint32_t* __restrict__ hist, uint32_t* __restrict__ offs, uint32_t* __restrict__ sorted, const uint8_t* __restrict__ skip) {
  if (skip && skip[blockIdx.x] == 2) {                   // all-zero row: every bucket empty
    for (int j = threadIdx.x; j < g.nb; j += blockDim.x) { hist[(size_t)blockIdx.x * g.nb + j] = 0; offs[(size_t)blockIdx.x * g.nb + j] = 0; }
    return;
  }
  uint32_t* cnt = sort_lds;                                     // nb counters, then cursors
  uint8_t* rng = reinterpret_cast<uint8_t*>(sort_lds + g.nb);   // range id of every bucket (start offset >> SORT_SL_LOG; the host keeps E <= 8*SORT_SL)
  uint32_t* part = sort_lds + g.nb + g.nb / 4;                  // 1024 scan partials
  uint32_t* rstart = part + 1024;                               // first position of every staged range
  uint32_t* stage = rstart + SORT_MAXR + 2;                     // SORT_SL + SORT_SLACK staged entries
  const size_t p = blockIdx.x;
  const int t = threadIdx.x, T = blockDim.x;
  const uint32_t ncol = (uint32_t)g.ncol, W = (uint32_t)(g.E / g.ncol);
  const dig_t* d = dig + p * g.E;
  uint32_t* out = sorted + p * g.estride;
  for (int j = t; j < g.nb; j += T) cnt[j] = 0;
  __syncthreads();
  for (uint32_t e = t; e < (uint32_t)g.E; e += T) { const int v = d[e]; if (v) atomicAdd(&cnt[(v < 0 ? -v : v) - 1], 1u); }
  __syncthreads();
  // block exclusive scan of the counts
  const int per = (g.nb + T - 1) / T;
  uint32_t sum = 0;
  for (int j = 0; j < per; j++) { const int b = t * per + j; if (b < g.nb) sum += cnt[b]; }
  part[t] = sum;
  __syncthreads();
  for (int o = 1; o < T; o <<= 1) { const uint32_t v = (t >= o) ? part[t - o] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
  const uint32_t total = part[T - 1];
  if (t <= (int)SORT_MAXR) rstart[t] = total;
  __syncthreads();
  uint32_t run = part[t] - sum;
  for (int j = 0; j < per; j++) {
    const int b = t * per + j;
    if (b < g.nb) {
      const uint32_t c = cnt[b];
      hist[p * g.nb + b] = c; offs[p * g.nb + b] = run;
      const uint32_t r = run >> SORT_SL_LOG;
      rng[b] = (uint8_t)(r < 255 ? r : 255);
      if (c && r < SORT_MAXR) atomicMin(&rstart[r], run);
      cnt[b] = run;                                             // becomes the cursor
      run += c;
    }
  }
  __syncthreads();
  const uint32_t nr = (total + SORT_SL - 1) >> SORT_SL_LOG;
  for (uint32_t r = 0; r < nr; r++) {
    const bool staged = r < SORT_MAXR;
    const uint32_t r0 = r << SORT_SL_LOG;
    const uint8_t rtag = (uint8_t)(r < 255 ? r : 255);
    for (uint32_t w = 0; w < W; w++) {
      const dig_t* dw = d + (size_t)w * ncol;
      const uint32_t tb = w * (uint32_t)g.tstride;
      for (uint32_t col = t; col < ncol; col += T) {
        const int v = dw[col];
        if (v == 0) continue;
        const int b = (v < 0 ? -v : v) - 1;
        if (rng[b] != rtag) continue;
        const uint32_t pos = atomicAdd(&cnt[b], 1u);
        const uint32_t ent = (tb + col) | (v < 0 ? 0x80000000u : 0u);
        const uint32_t idx = pos - r0;
        if (staged && idx < SORT_SL + SORT_SLACK) stage[idx] = ent; else out[pos] = ent;
      }
    }
    __syncthreads();
    if (staged) {
      uint32_t lo = rstart[r], hi = total;
      for (uint32_t q = r + 1; q < nr && q <= SORT_MAXR; q++) if (rstart[q] < total) { hi = rstart[q]; break; }
      if (lo < total) {
        const uint32_t cap = r0 + SORT_SL + SORT_SLACK;
        if (hi > cap) hi = cap;                                 // the overhang past the slack went straight to global memory
        for (uint32_t pos = lo + t; pos < hi; pos += T) out[pos] = stage[pos - r0];
      }
    }
    __syncthreads();
  }
}

// exclusive scan per problem: offs[p][b] = sum_{b'<b} hist[p][b'].  One block per problem.
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ hist, uint32_t* __restrict__ offs, int nb) {
  __shared__ uint32_t part[1024];
  const size_t w = blockIdx.x; const int t = threadIdx.x, T = blockDim.x;
  const int per = (nb + T - 1) / T;
  const uint32_t* h = hist + w * nb;
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
    if (b < nb) { offs[w * nb + b] = run; run += h[b]; }
  }
}

// ---- bucket accumulation -------------------------------------------------------------------------
// A bucket's sorted entry list is cut into segments of SEG entries.  The owner lane takes segment 0; a bucket with
// more (skewed scalars, or the short top window of a single MSM whose few buckets hold n/2^tb points each) posts its
// other segments to a work list that k_acc_extra spreads over the whole chip, and k_acc_merge folds the partials
// with one wave per oversized bucket.  No lane ever runs a chain longer than SEG mixed adds (+ a short merge).
constexpr uint32_t ACC_SEG_MAX = 8192;   // longest chain one lane runs before a bucket is cut into segments
struct AccCounters { uint32_t extra_count, big_count; };
struct ExtraItem { uint32_t bucket, seg; };
struct BigItem { uint32_t bucket, base, k; };

// software-pipelined: the index two entries ahead and the point one entry ahead are in flight during each mixed add
__device__ __forceinline__ XYZZ acc_segment(const uint32_t* __restrict__ points, const uint32_t* __restrict__ lst, uint32_t from, uint32_t to) {
  XYZZ acc = xyzz_inf();
  if (from >= to) return acc;
  uint32_t e_cur = lst[from];
  uint32_t e_nxt = (from + 1 < to) ? lst[from + 1] : 0;
  Affine p_cur = aff_load(points + 16 * (size_t)(e_cur & 0x7fffffffu));
  for (uint32_t j = from; j < to; j++) {
    const uint32_t e = e_cur; const Affine p = p_cur;
    if (j + 1 < to) {
      e_cur = e_nxt;
      p_cur = aff_load(points + 16 * (size_t)(e_cur & 0x7fffffffu));
      if (j + 2 < to) e_nxt = lst[j + 2];
    }
    xyzz_madd(acc, p, (e >> 31) != 0);
  }
  return acc;
}

// Buckets are processed in order of decreasing load (global counting sort on min(count, SEG)): the 64 lanes of a wave
// then run chains of equal length instead of waiting for the longest of 64 Poisson draws, and the heaviest waves are
// dispatched first so the tail of the launch is made of the lightest ones.
__global__ void __launch_bounds__(1024) k_size_hist(const uint32_t* __restrict__ hist, size_t nbuckets, uint32_t SEG, uint32_t* __restrict__ gbins) {
  __shared__ uint32_t bins[ACC_SEG_MAX + 2];
  for (uint32_t j = threadIdx.x; j <= SEG; j += blockDim.x) bins[j] = 0;
  __syncthreads();
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nbuckets) { const uint32_t cn = hist[t]; atomicAdd(&bins[SEG - (cn < SEG ? cn : SEG)], 1u); }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j <= SEG; j += blockDim.x) if (bins[j]) atomicAdd(&gbins[j], bins[j]);
}
// exclusive scan of the SEG + 1 bins by one wave: lane l owns a contiguous run of bins, the runs' totals are scanned with shuffles
// (one lane walking all bins was a 27 us dependent chain on the critical path of a single MSM)
__global__ void __launch_bounds__(64) k_size_scan(uint32_t* __restrict__ gbins, uint32_t SEG) {
  const uint32_t n = SEG + 1, per = (n + 63) / 64, lane = threadIdx.x;
  const uint32_t j0 = lane * per, j1 = (j0 + per < n) ? j0 + per : n;
  uint32_t sum = 0;
  for (uint32_t j = j0; j < j1; j++) sum += gbins[j];
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += o; }
  uint32_t run = incl - sum;
  for (uint32_t j = j0; j < j1; j++) { const uint32_t v = gbins[j]; gbins[j] = run; run += v; }
}
__global__ void __launch_bounds__(1024) k_size_scatter(const uint32_t* __restrict__ hist, size_t nbuckets, uint32_t SEG, uint32_t* __restrict__ gcur, uint32_t* __restrict__ perm) {
  __shared__ uint32_t bins[ACC_SEG_MAX + 2];
  __shared__ uint32_t base[ACC_SEG_MAX + 2];
  for (uint32_t j = threadIdx.x; j <= SEG; j += blockDim.x) bins[j] = 0;
  __syncthreads();
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t key = 0, rank = 0;
  if (t < nbuckets) { const uint32_t cn = hist[t]; key = SEG - (cn < SEG ? cn : SEG); rank = atomicAdd(&bins[key], 1u); }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j <= SEG; j += blockDim.x) if (bins[j]) base[j] = atomicAdd(&gcur[j], bins[j]);
  __syncthreads();
  if (t < nbuckets) perm[base[key] + rank] = (uint32_t)t;
}

// G lanes per bucket: each takes 1/G of the bucket's first SEG entries and stores its own partial sum (bucket t owns slots
// t*G .. t*G+G-1); the reduction adds the G partials as it loads a bucket.  Shorter chains for the same work: the launch ramps
// down on chains of cnt/G additions instead of cnt, and the short top window's double-length buckets stop being the critical
// path — without touching the accumulate kernel's register budget (a fold inside this kernel costs a wave per SIMD).
template <int G>
__global__ void __launch_bounds__(256) k_acc_first(const uint32_t* __restrict__ points, size_t nbuckets, int nb, size_t estride, uint32_t SEG,
                                                   const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ sorted,
                                                   const uint32_t* __restrict__ perm,
                                                   uint32_t* __restrict__ buckets, AccCounters* __restrict__ ctr, ExtraItem* __restrict__ extra, BigItem* __restrict__ big) {
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t t0 = gl / G; const uint32_t part = (uint32_t)(gl % G);
  const bool live = t0 < nbuckets;
  const size_t t = live ? perm[t0] : 0;
  const uint32_t cnt = live ? hist[t] : 0;
  const uint32_t* lst = sorted + (t / nb) * estride + (live ? offs[t] : 0);
  const uint32_t first = cnt < SEG ? cnt : SEG;
  const uint32_t per = (first + G - 1) / G;
  uint32_t from = part * per, to = from + per; if (from > first) from = first; if (to > first) to = first;
  const XYZZ acc = acc_segment(points, lst, from, to);
  if (!live) return;
  xyzz_store(buckets + 32 * (t * G + part), acc);        // G partial sums per bucket, folded by k_reduce_l1 as it loads them
  if (part) return;
  if (cnt > SEG) {
    const uint32_t k = (cnt + SEG - 1) / SEG - 1;
    const uint32_t base = atomicAdd(&ctr->extra_count, k);
    for (uint32_t j = 0; j < k; j++) { ExtraItem it; it.bucket = (uint32_t)t; it.seg = j + 1; extra[base + j] = it; }
    BigItem bi; bi.bucket = (uint32_t)t; bi.base = base; bi.k = k;
    big[atomicAdd(&ctr->big_count, 1u)] = bi;
  }
}
__global__ void __launch_bounds__(256) k_acc_extra(const uint32_t* __restrict__ points, int nb, size_t estride, uint32_t SEG,
                                                   const uint32_t* __restrict__ hist, const uint32_t* __restrict__ offs, const uint32_t* __restrict__ sorted,
                                                   const AccCounters* __restrict__ ctr, const ExtraItem* __restrict__ extra, uint32_t* __restrict__ extra_out) {
  const uint32_t total = ctr->extra_count;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const ExtraItem it = extra[i];
    const size_t t = it.bucket;
    const uint32_t cnt = hist[t];
    const uint32_t* lst = sorted + (t / nb) * estride + offs[t];
    const uint32_t from = it.seg * SEG, to = (from + SEG < cnt) ? from + SEG : cnt;
    xyzz_store(extra_out + 32 * i, acc_segment(points, lst, from, to));
  }
}

// ---- weighted bucket sums --------------------------------------------------------------------
// wave-level helpers: every lane holds one point
__device__ __forceinline__ XYZZ wave_suffix_scan(XYZZ v, int lane) {   // v_l <- sum_{l' >= l} v_l'
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    XYZZ o = xyzz_shfl_down(v, d);
    if (lane + d < 64) v = xyzz_add_inl(v, o);
  }
  return v;
}
__device__ __forceinline__ XYZZ wave_sum(XYZZ v, int lane) {           // lane 0 <- sum of all lanes
#pragma unroll 1
  for (int d = 32; d >= 1; d >>= 1) {
    XYZZ o = xyzz_shfl_down(v, d);
    if (lane < d) v = xyzz_add_inl(v, o);
  }
  return v;
}
__device__ __forceinline__ XYZZ xyzz_mul_pow2(XYZZ v, int k) {
#pragma unroll 1
  for (int i = 0; i < k; i++) v = xyzz_dbl(v);
  return v;
}

// k * v for a small positive integer k (binary double-and-add from the top bit)
__device__ __forceinline__ XYZZ xyzz_mul_small(const XYZZ& v, uint32_t k) {
  if (k <= 1) return v;
  int top = 31 - __clz((int)k);
  XYZZ r = v;
#pragma unroll 1
  for (int b = top - 1; b >= 0; b--) {
    r = xyzz_dbl(r);
    if ((k >> b) & 1u) r = xyzz_add_inl(r, v);
  }
  return r;
}

// folding the extra partials back into their bucket: a bucket with few of them is finished by one lane (64 buckets per
// wave), a bucket with many by one whole wave (lane-strided chains, then a wave tree)
constexpr uint32_t MERGE_LANE_MAX = 12;
// (`G` = slots per bucket: the extras are folded into the bucket's first slot)
__global__ void __launch_bounds__(64) k_acc_merge(const AccCounters* __restrict__ ctr, const BigItem* __restrict__ big, const uint32_t* __restrict__ extra_out, uint32_t* __restrict__ buckets, int G) {
  const int lane = threadIdx.x;
  const uint32_t total = ctr->big_count;
  if (total == 0) return;                                  // uniform scalars: no oversized bucket — the launch is a no-op (one launch, not two, since round 4)
  // buckets with few partials: one lane each
  for (size_t i = (size_t)blockIdx.x * 64 + lane; i < total; i += (size_t)gridDim.x * 64) {
    const BigItem bi = big[i];
    if (bi.k > MERGE_LANE_MAX) continue;
    XYZZ acc = xyzz_load(buckets + 32 * (size_t)bi.bucket * G);
    for (uint32_t j = 0; j < bi.k; j++) acc = xyzz_add(acc, xyzz_load(extra_out + 32 * (size_t)(bi.base + j)));
    xyzz_store(buckets + 32 * (size_t)bi.bucket * G, acc);
  }
  // buckets with many: one wave each (different buckets than above: no ordering between the two loops is needed)
  for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {
    const BigItem bi = big[i];
    if (bi.k <= MERGE_LANE_MAX) continue;
    XYZZ acc = (lane == 0) ? xyzz_load(buckets + 32 * (size_t)bi.bucket * G) : xyzz_inf();
    for (uint32_t j = lane; j < bi.k; j += 64) acc = xyzz_add_inl(acc, xyzz_load(extra_out + 32 * (size_t)(bi.base + j)));
    acc = wave_sum_all(acc);
    if (lane == 0) xyzz_store(buckets + 32 * (size_t)bi.bucket * G, acc);
  }
}

// level 1: one wave per chunk of 64*L consecutive buckets of one problem (window).
//   S  = sum_i X_i ,  Wt = sum_i i * X_i   (i = 0-based index inside the chunk)
// out[chunk] = {S, Wt}
// G = partial sums per bucket (k_acc_first<G>): bucket i of the chunk is the sum of slots i*G .. i*G+G-1.
__device__ __forceinline__ XYZZ bucket_load(const uint32_t* __restrict__ p, int G) {
  XYZZ x = xyzz_load(p);
  for (int g = 1; g < G; g++) x = xyzz_add_inl(x, xyzz_load(p + 32 * g));
  return x;
}
// L: buckets per lane (any value >= 1: the host picks it so that the chunks spread evenly over the SIMDs); nb: buckets per problem; the last chunk of a
// problem may be ragged (buckets past nb count as the identity).
__global__ void __launch_bounds__(64) k_reduce_l1(const uint32_t* __restrict__ X, int L, int nb, uint32_t* __restrict__ out, const uint8_t* __restrict__ skip, int chunks_per_problem, int G) {
  const int lane = threadIdx.x;
  const size_t chunk = blockIdx.x;
  const size_t prob = chunk / chunks_per_problem; const int cidx = (int)(chunk % chunks_per_problem);
  if (skip && skip[prob] == 2) {   // all-zero row: nothing was accumulated
    if (lane == 0) { xyzz_store(out + 64 * chunk, xyzz_inf()); xyzz_store(out + 64 * chunk + 32, xyzz_inf()); }
    return;
  }
  const int b0 = (cidx * 64 + lane) * L;                       // this lane's first bucket inside the problem
  const uint32_t* base = X + 32 * (prob * (size_t)nb + b0) * G;
  // lane-sequential running sums over its L buckets: run = sum X_i, acc = sum i*X_i (local i)
  XYZZ run = xyzz_inf(), acc = xyzz_inf();
  for (int i = L - 1; i >= 1; i--) {
    if (b0 + i < nb) run = xyzz_add_inl(run, bucket_load(base + 32 * (size_t)i * G, G));
    acc = xyzz_add_inl(acc, run);
  }
  if (b0 < nb) run = xyzz_add_inl(run, bucket_load(base, G));
  // across lanes: index = lane*L + local  =>  Wt = sum_l acc_l + L * sum_l l * run_l
  XYZZ suf = wave_suffix_scan(run, lane);          // suf_l = sum_{l'>=l} run_l'
  XYZZ S = suf;                                    // lane 0 holds the chunk total
  XYZZ term = (lane >= 1) ? suf : xyzz_inf();      // sum_{l>=1} suf_l = sum_l l*run_l
  term = xyzz_mul_small(term, (uint32_t)L);
  term = xyzz_add_inl(term, acc);
  const XYZZ Wt = wave_sum_all(term);              // 7 quad-cooperative additions deep (g1.cuh), every lane gets the sum
  if (lane == 0) { xyzz_store(out + 64 * chunk, S); xyzz_store(out + 64 * chunk + 32, Wt); }
}

// combine: one wave per group of 64 consecutive chunk results of one problem; lane l holds chunk g0+l, each chunk
// standing for M = Lb * 64^k64 buckets (Lb: buckets per lane of level 1, any value).  Produces the {S, Wt} of the 64*M-bucket super-chunk:
//   S' = sum_l S_l ,  Wt' = sum_l Wt_l + M * sum_l l*S_l
// G = chunks per problem at the input level (lanes past G contribute the identity); Gout = ceil(G/64).
// When `final` is set (Gout == 1) the wave writes sum_b (b+1) X_b = Wt' + S' as a single point instead.
__global__ void __launch_bounds__(64) k_reduce_combine(const uint32_t* __restrict__ in, int G, int Gout, int k64, int Lb, int final, uint32_t* __restrict__ out) {
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
  term = xyzz_mul_small(xyzz_mul_pow2(term, 6 * k64), (uint32_t)Lb);
  term = xyzz_add_inl(term, Wt);
  const XYZZ W2 = wave_sum_all(term);
  if (lane == 0) {
    if (final) {
      xyzz_store(out + 32 * prob, xyzz_add_inl(W2, suf));
    } else {
      uint32_t* o = out + 64 * (prob * (size_t)Gout + grp);
      xyzz_store(o, suf); xyzz_store(o + 32, W2);
    }
  }
}

// ---- the combine level for the LATENCY regime (few chunks: a single MSM of ~2^20 points, small commits) ----------------
// One wave per group leaves the chip idle and runs a chain of full additions (14 dependent products, 7.6 us each on a lone wave).
// Here a group is a block of 256 threads = 64 QUADS, each quad standing for one lane of the kernel above: every addition is the
// quad-cooperative one (g1.cuh: 4 dependent products, ~3.5 us), all four lanes of a quad hold the same values, and quads exchange
// points through LDS (64 x 128 B).  Same sums, same order of operands inside every addition as far as the group law cares (the
// results are group elements: any order gives the same point).
__device__ __forceinline__ void quad_lds_put(uint32_t* sm, int q, int role, const XYZZ& v) { if (role == 0) xyzz_store(sm + 32 * q, v); }
// suffix scan over the 64 quads of the block: v_q <- sum_{q' >= q} v_q'
__device__ __forceinline__ XYZZ quad_suffix_scan(XYZZ v, int q, int role, uint32_t* sm) {
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    quad_lds_put(sm, q, role, v);
    __syncthreads();
    const XYZZ o = (q + d < 64) ? xyzz_load(sm + 32 * (q + d)) : xyzz_inf();
    __syncthreads();
    if (q + d < 64) v = xyzz_add_quad(v, o, role);
  }
  return v;
}
// sum over the 64 quads, returned in quad 0
__device__ __forceinline__ XYZZ quad_tree_sum(XYZZ v, int q, int role, uint32_t* sm) {
#pragma unroll 1
  for (int d = 32; d >= 1; d >>= 1) {
    quad_lds_put(sm, q, role, v);
    __syncthreads();
    const XYZZ o = (q < d) ? xyzz_load(sm + 32 * (q + d)) : xyzz_inf();
    __syncthreads();
    if (q < d) v = xyzz_add_quad(v, o, role);
  }
  return v;
}
__global__ void __launch_bounds__(256) k_reduce_combine_quad(const uint32_t* __restrict__ in, int G, int Gout, int k64, int Lb, int final, uint32_t* __restrict__ out) {
  __shared__ __align__(16) uint32_t sm[64 * 32];
  const int q = threadIdx.x >> 2, role = threadIdx.x & 3;
  const size_t prob = blockIdx.x / Gout;
  const int grp = blockIdx.x % Gout;
  const int g = grp * 64 + q;
  XYZZ S = xyzz_inf(), Wt = xyzz_inf();
  if (g < G) {
    const uint32_t* p = in + 64 * (prob * (size_t)G + g);
    S = xyzz_load(p); Wt = xyzz_load(p + 32);
  }
  const XYZZ suf = quad_suffix_scan(S, q, role, sm);
  XYZZ term = (q >= 1) ? suf : xyzz_inf();
  term = xyzz_mul_small(xyzz_mul_pow2(term, 6 * k64), (uint32_t)Lb);
  term = xyzz_add_quad(term, Wt, role);
  const XYZZ W2 = quad_tree_sum(term, q, role, sm);
  if (threadIdx.x == 0) {
    if (final) {
      xyzz_store(out + 32 * prob, xyzz_add_inl(W2, suf));
    } else {
      uint32_t* o = out + 64 * (prob * (size_t)Gout + grp);
      xyzz_store(o, suf); xyzz_store(o + 32, W2);
    }
  }
}

// window table for MODE_ROWS: slab w holds 2^(c*w) * P_j for every base j (XYZZ here; k_xyzz_to_affine finishes it).
// Setup only, once per generator set and window size.
__global__ void __launch_bounds__(64) k_window_table(const uint32_t* __restrict__ bases /* npts Montgomery affine */, size_t npts, int c, int W, uint32_t* __restrict__ out_xyzz) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npts) return;
  XYZZ p = xyzz_from_affine(aff_load(bases + 16 * j));
  for (int w = 0; w < W; w++) {
    xyzz_store(out_xyzz + 32 * ((size_t)w * npts + j), p);
    if (w + 1 < W) p = xyzz_mul_pow2(p, c);
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
  if (out_xy) { fe_store_packed<FqP>(out_xy + 16 * i, fe_from_mont(a.x)); fe_store_packed<FqP>(out_xy + 16 * i + 8, fe_from_mont(a.y)); }
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

// ---- merging the scalars of equal bases ---------------------------------------------------------
// The reference derives its generators as s*G with s = 1 whenever SHA3(bytes) >= r (group.rs:110-131), so ~66 % of a
// generator set is literally the point G.  sum_j k_j P_j = sum_u (sum_{j: P_j = U_u} k_j) U_u: the scalars of equal bases
// are added mod r first (32 B streamed per scalar, one Fr add) and the MSM runs over the unique bases only.
// csr_off[u]..csr_off[u+1] lists the columns of unique base u; column == R means the blind column (base h).
__device__ __forceinline__ Fr merged_load(const uint32_t* __restrict__ Z, const uint32_t* __restrict__ blinds, size_t row, size_t R, uint32_t col) {
  if (col < R) return fe_load<FrP>(Z + 8 * (row * R + col));
  if (blinds) return fe_load<FrP>(blinds + 8 * row);
  return fe_zero<FrP>();
}
// Rows whose R scalars are all equal (the padded tail of every derefs matrix repeats mem[0]: sparse_mlpoly_full.rs:89-101,
// ~43 % of the non-zero rows at keyless size) commit to v * (sum_j G_j): the unique-base table carries that sum as one
// extra column, the row's value goes there and every other merged scalar of the row is zero (only h keeps the blind).
// flags[row] = 0: general row, 1: constant non-zero row, 2: all-zero row (hyrax.rs:245 padding; skipped by the later
// stages when there are no blinds).  One block per row.
// This is also the pass that sees every input scalar of the merge path: values >= r are counted in *bad (the Fr additions
// that follow assume canonical operands; Scalar::from_bytes rejects them, scalar.rs:87-95).
__global__ void __launch_bounds__(256) k_row_const_flags(const uint32_t* __restrict__ Z, const uint32_t* __restrict__ blinds, size_t R, uint8_t* __restrict__ flags, uint32_t* __restrict__ bad) {
  const size_t row = blockIdx.x;
  const uint4* z = reinterpret_cast<const uint4*>(Z + 8 * row * R);
  const uint4 a0 = z[0], a1 = z[1];
  int diff = 0, nbad = 0;
  for (size_t j = threadIdx.x; j < R; j += blockDim.x) {
    const uint4 b0 = z[2 * j], b1 = z[2 * j + 1];
    diff |= (a0.x ^ b0.x) | (a0.y ^ b0.y) | (a0.z ^ b0.z) | (a0.w ^ b0.w) | (a1.x ^ b1.x) | (a1.y ^ b1.y) | (a1.z ^ b1.z) | (a1.w ^ b1.w);
    const uint32_t kk[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    nbad += (!bad || fe_is_canonical<FrP>(kk)) ? 0 : 1;          // bad == nullptr: table values (lazy Montgomery representatives), nothing to check
  }
  if (blinds && threadIdx.x == 0) {
    const uint4 b0 = reinterpret_cast<const uint4*>(blinds + 8 * row)[0], b1 = reinterpret_cast<const uint4*>(blinds + 8 * row)[1];
    const uint32_t kk[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    nbad += fe_is_canonical<FrP>(kk) ? 0 : 1;
  }
  if (nbad && bad) atomicAdd(bad, (uint32_t)nbad);
  const int any = __syncthreads_or(diff != 0);
  if (threadIdx.x == 0) flags[row] = any ? 0 : (((a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w) == 0) ? 2 : 1);
}
// small groups: one lane per (row, merged column).  Column U is the sum-of-all-bases column; hcol = the unique base h maps to
// (or U+1 when the table has no h).
__device__ __forceinline__ void merge_small_body(size_t bidx, const uint32_t* __restrict__ Z, const uint32_t* __restrict__ blinds, size_t L, size_t R, size_t U,
                                                 const uint32_t* __restrict__ csr_off, const uint32_t* __restrict__ csr_cols, uint32_t big_threshold,
                                                 const uint8_t* __restrict__ flags, uint32_t hcol, uint32_t* __restrict__ out, int internal) {
  const size_t t = bidx * blockDim.x + threadIdx.x;
  const size_t W1 = U + 1;
  if (t >= L * W1) return;
  const size_t row = t / W1, u = t - row * W1;
  const bool cst = flags && flags[row] != 0;            // flags == nullptr: rows written by this library (bullet rounds), never constant
  // internal != 0: Z (and the blinds) are this library's TABLE values (Montgomery R = 2^261, lazy representatives below 2.5 r: sbn_commit_table) —
  // the sums are taken on them and leave Montgomery form here, on the way out: the merge reads every scalar anyway, so the separate
  // conversion pass over the matrix (1 GiB read + 1 GiB written for the derefs commitment) disappears.
  // (copies of canonical input scalars are stored as they are; sums go through fe_store, which canonicalises: the digits are cut from them)
  if (u == U) { const Fr v = cst ? fe_load<FrP>(Z + 8 * row * R) : fe_zero<FrP>(); fe_store_packed<FrP>(out + 8 * t, internal ? fe_from_mont(v) : v); return; }
  if (cst) { const Fr v = (u == hcol && blinds) ? fe_load<FrP>(blinds + 8 * row) : fe_zero<FrP>(); fe_store_packed<FrP>(out + 8 * t, internal ? fe_from_mont(v) : v); return; }
  const uint32_t a = csr_off[u], b = csr_off[u + 1];
  if (b - a > big_threshold) return;                     // k_merge_big owns it
  Fr acc = merged_load(Z, blinds, row, R, csr_cols[a]);
  for (uint32_t j = a + 1; j < b; j++) acc = fe_add(acc, merged_load(Z, blinds, row, R, csr_cols[j]));      // <= MERGE_BIG terms (160 r at most)
  if (internal) fe_store_packed<FrP>(out + 8 * t, fe_from_mont(fe_reduce(acc))); else fe_store<FrP>(out + 8 * t, acc);
}
// big groups: one block per (row, big group), four scalars in flight per lane (a bullet round merges 2 x 5381 scalars: one wave
// with one load at a time was a chain of 84 exposed load latencies); constant rows were written by k_merge_small
__device__ __forceinline__ void merge_big_body(size_t bidx, const uint32_t* __restrict__ Z, const uint32_t* __restrict__ blinds, size_t L, size_t R, size_t U,
                                               const uint32_t* __restrict__ csr_off, const uint32_t* __restrict__ csr_cols,
                                               const uint32_t* __restrict__ big_list, uint32_t nbig, const uint8_t* __restrict__ flags, uint32_t* __restrict__ out, int internal) {
  __shared__ uint32_t sm[4][NL];
  const size_t row = bidx / nbig; const uint32_t u = big_list[bidx % nbig];
  if (flags && flags[row]) return;
  const uint32_t a = csr_off[u], b = csr_off[u + 1];
  Fr acc = fe_zero<FrP>();
  uint32_t cnt = 0;
  for (uint32_t j = a + threadIdx.x; j < b; j += 1024) {
    const Fr x0 = merged_load(Z, blinds, row, R, csr_cols[j]);
    const Fr x1 = j + 256 < b ? merged_load(Z, blinds, row, R, csr_cols[j + 256]) : fe_zero<FrP>();
    const Fr x2 = j + 512 < b ? merged_load(Z, blinds, row, R, csr_cols[j + 512]) : fe_zero<FrP>();
    const Fr x3 = j + 768 < b ? merged_load(Z, blinds, row, R, csr_cols[j + 768]) : fe_zero<FrP>();
    acc = fe_add(fe_add(acc, x0), fe_add(fe_add(x1, x2), x3));
    if (((cnt += 4) & (internal ? 31u : 127u)) == 0) acc = fe_reduce(acc);   // keep the running sum far below the top limb's range (table values reach 2.5 r: every 32 terms)
  }
  acc = fe_reduce(acc);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc = fe_add(acc, fe_shfl_down(acc, d));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { for (int k = 0; k < NL; k++) sm[wv][k] = acc.v[k]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    Fr s = acc;
    for (int w = 1; w < 4; w++) { Fr x; for (int k = 0; k < NL; k++) x.v[k] = sm[w][k]; s = fe_add(s, x); }
    if (internal) fe_store_packed<FrP>(out + 8 * (row * (U + 1) + u), fe_from_mont(fe_reduce(s))); else fe_store<FrP>(out + 8 * (row * (U + 1) + u), s);
  }
}
// both kinds of merged column in ONE launch: blocks [0, nsmall) walk the (row, column) pairs of the small groups, the rest take one
// (row, big group) each — a bullet round's merge was two back-to-back launches of 8 + 15 us
__global__ void __launch_bounds__(256) k_merge(const uint32_t* __restrict__ Z, const uint32_t* __restrict__ blinds, size_t L, size_t R, size_t U,
                                               const uint32_t* __restrict__ csr_off, const uint32_t* __restrict__ csr_cols, uint32_t big_threshold,
                                               const uint32_t* __restrict__ big_list, uint32_t nbig, const uint8_t* __restrict__ flags, uint32_t hcol, uint32_t* __restrict__ out, int internal,
                                               uint32_t nsmall) {
  if (blockIdx.x < nsmall) merge_small_body(blockIdx.x, Z, blinds, L, R, U, csr_off, csr_cols, big_threshold, flags, hcol, out, internal);
  else merge_big_body(blockIdx.x - nsmall, Z, blinds, L, R, U, csr_off, csr_cols, big_list, nbig, flags, out, internal);
}

// Jacobian (X, Y, Z) -> Montgomery affine, one lane per point (x = X/Z^2, y = Y/Z^3); canonical input is converted first
__global__ void __launch_bounds__(64) k_jacobian_to_affine(const uint32_t* __restrict__ in, int in_is_mont, size_t n, uint32_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fq X = fe_load<FqP>(in + 24 * i), Y = fe_load<FqP>(in + 24 * i + 8), Z = fe_load<FqP>(in + 24 * i + 16);
  if (!in_is_mont) { X = fe_to_mont(X); Y = fe_to_mont(Y); Z = fe_to_mont(Z); }
  else { X = fe_from_ark_mont(X); Y = fe_from_ark_mont(Y); Z = fe_from_ark_mont(Z); }
  Affine a;
  if (fe_is_zero(Z)) { a.x = fe_zero<FqP>(); a.y = a.x; }
  else { const Fq zi = fe_inv(Z), zi2 = fe_sqr(zi); a.x = fe_mul(X, zi2); a.y = fe_mul(Y, fe_mul(zi2, zi)); }
  aff_store(out + 16 * i, a);
}
// s * P_i for every point of a table (MultiCommitGens::scale): double-and-add, scalar shared by all lanes
__global__ void __launch_bounds__(64) k_scale_points(const uint32_t* __restrict__ pts, size_t n, const uint32_t* __restrict__ scalar /* canonical */, uint32_t* __restrict__ out_xyzz) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine P = aff_load(pts + 16 * i);
  XYZZ acc = xyzz_inf();
  int top = 255;
  while (top >= 0 && !((scalar[top >> 5] >> (top & 31)) & 1u)) top--;
  for (int b = top; b >= 0; b--) {
    acc = xyzz_dbl(acc);
    if ((scalar[b >> 5] >> (b & 31)) & 1u) xyzz_madd(acc, P, false);
  }
  xyzz_store(out_xyzz + 32 * i, acc);
}

// P_i = P0 + (first + i) * D : lane-local double-and-add on the 64-bit index (synthetic bases with known dlogs)
__global__ void __launch_bounds__(64) k_arith_points(const uint32_t* __restrict__ p0d /* P0, D as XYZZ */, unsigned long long first, size_t n, uint32_t* __restrict__ out_xyzz) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const XYZZ P0 = xyzz_load(p0d), D = xyzz_load(p0d + 32);
  unsigned long long k = first + i;
  XYZZ acc = xyzz_inf();
  for (int b = 63; b >= 0; b--) {
    if (!xyzz_is_inf(acc)) acc = xyzz_dbl(acc);
    if ((k >> b) & 1ull) acc = xyzz_add(acc, D);
  }
  xyzz_store(out_xyzz + 32 * i, xyzz_add(acc, P0));
}
// Synthetic benchmark scalars: scalar t = the four SplitMix64 outputs of counters 4(first+t)+1 .. +4 as
// little-endian u64 limbs, top limb cut to 62 bits, minus r when >= r (a 254-bit value is < 2r) -> canonical, full width.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_scalars_synthetic(unsigned long long seed, unsigned long long first, size_t n, uint32_t* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint32_t k[8];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    unsigned long long z = splitmix64(seed, 4ull * (first + t) + (unsigned long long)j + 1ull);
    if (j == 3) z &= (1ull << 62) - 1ull;
    k[2 * j] = (uint32_t)z; k[2 * j + 1] = (uint32_t)(z >> 32);
  }
  if (!fe_is_canonical<FrP>(k)) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)k[i] - modlimb<FrP>(i) - borrow; k[i] = (uint32_t)d; borrow = d >> 63; }
  }
  uint4* o = reinterpret_cast<uint4*>(out + 8 * t);
  o[0] = make_uint4(k[0], k[1], k[2], k[3]); o[1] = make_uint4(k[4], k[5], k[6], k[7]);
}
__global__ void __launch_bounds__(256) k_points_from_mont(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe_store_packed<FqP>(out + 16 * i, fe_from_mont(fe_load<FqP>(in + 16 * i)));
  fe_store_packed<FqP>(out + 16 * i + 8, fe_from_mont(fe_load<FqP>(in + 16 * i + 8)));
}

}  // namespace sbn
