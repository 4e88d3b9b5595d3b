// sort2_kernels.cuh — two-level counting sort of the digits of ONE large MSM (group.rs:171-175 at 2^22 .. 2^26 terms).
//
// The one-level sort (msm_kernels.cuh: k_hist_lds / k_scatter_lds) writes every (point index, sign) entry straight to its final
// slot: 4-byte stores spread over a window's whole sorted array (256 MB at 2^26), which the memory system serves as one
// read-modify-write of a 64-byte sector each — 28 of the 125 ms of a 2^26 MSM — and it needs all 2^(c-1) counters of a window in
// LDS, which caps the window at c = 16.  Here the bucket index b = hi * 2048 + lo is sorted in two steps:
//   level 1  (k_s2_count, k_s2_prefix_*, k_s2_scatter)  a block takes 8192 scalars, recodes them (no digit array: the digits
//            are recomputed from the scalars, 32 B each, instead of being written once and read W times), sorts its
//            entries of one window by hi inside LDS and writes them out as runs: (index | sign) and lo, partitioned by hi.
//   level 2  (k_s2_hist, k_s2_prefix2, k_s2_place)  a block takes 16384 entries of ONE partition: LDS counters for its <= 2048
//            buckets; the entries are sorted by lo inside LDS and leave as runs into the partition's own slice of the sorted array.
// Every later stage (k_size_*, k_acc_*, k_reduce_*) sees the same hist / offs / sorted arrays as after the one-level sort, so
// nothing downstream changes — except that c may now exceed 16 (13 windows instead of 16 at 2^26).
// The order of entries inside a bucket depends on LDS arbitration only; bucket sums do not.
#pragma once
#include "msm_kernels.cuh"
#include <utility>

namespace sbn {

constexpr int S2_LO_LOG_MAX = 11, S2_LO_MAX = 1 << S2_LO_LOG_MAX;      // most buckets per partition (LDS counters of level 2)
constexpr int S2_P_MAX = 1024;                                        // most partitions per window (one scan lane each in level 1)
constexpr int S2_SPT = 8, S2_CH = 1024 * S2_SPT;          // scalars per thread / per block in level 1 (large inputs)
constexpr int S2_SPT_SMALL = 2;                           // ... up to 2^21 scalars: 8192 per block would be 128 - 256 blocks, half the chip or less (k_s2_scatter at 2^20: 95 us)
constexpr int S2_C_MIN = 13, S2_C_MAX = 22;               // window bits this path is built for (P = 2^(c-12) partitions: 2 .. 1024)
struct S2Geom {
  size_t n;        // scalars
  int c, W, P;     // window bits, windows, partitions per window
  int lo_log;      // log2 of the buckets per partition: bucket b = hi << lo_log | lo
  int K;           // level-1 chunks (blocks)
};

// Digits are taken from scalars held in REGISTERS: window bits C and window index w are compile-time (the kernels are
// instantiated per C and unrolled over w), so limb and shift of every digit are constants and a scalar is read from memory once
// per pass (two 16-byte loads per lane) instead of two or three words per window with a 32-byte lane stride.
template <int C> struct S2Shape {
  static constexpr int W0 = (254 + C - 1) / C;
  static constexpr int W = W0 + ((254 - (W0 - 1) * C) > C - 1 ? 1 : 0);      // == make_shape(C).W
};
// (the scalar is a VALUE of vector type, not an array: an array gets its two-word reads merged into overlapping 64-bit loads,
//  which keeps it in scratch memory)
typedef uint32_t s2_u32x8 __attribute__((ext_vector_type(8)));
template <int C, int w> __device__ __forceinline__ int s2_digit(const s2_u32x8 k, uint32_t& carry) {
  constexpr int bit = w * C, limb = bit >> 5, sh = bit & 31;
  uint64_t x = 0;
  if constexpr (limb < 8) x = k[limb];
  if constexpr (limb + 1 < 8) x |= (uint64_t)k[limb + 1] << 32;
  const uint32_t raw = (uint32_t)(x >> sh) & ((1u << C) - 1u);
  const uint32_t v = raw + carry;
  if (v >= (1u << (C - 1))) { carry = 1; return (int)v - (1 << C); }
  carry = 0;
  return (int)v;
}
__device__ __forceinline__ s2_u32x8 s2_load_scalar(const uint32_t* __restrict__ p) {
  const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
  s2_u32x8 k; k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
  return k;
}
__device__ __forceinline__ bool s2_is_canonical(const s2_u32x8 k) { uint32_t kk[8]; for (int i = 0; i < 8; i++) kk[i] = k[i]; return fe_is_canonical<FrP>(kk); }

extern __shared__ uint32_t s2_lds[];

// level 1, pass A: per block and window, how many entries fall into each partition.  cntA[(w*P + hi)*K + k]
template <int C, int... Ws>
__device__ __forceinline__ void s2_count_scalar(const s2_u32x8 kk, int P, int lo_log, std::integer_sequence<int, Ws...>) {
  uint32_t carry = 0;
  auto one = [&](auto wc) {
    constexpr int w = decltype(wc)::value;
    const int d = s2_digit<C, w>(kk, carry);
    if (d != 0) atomicAdd(&s2_lds[w * P + (((d < 0 ? -d : d) - 1) >> lo_log)], 1u);
  };
  (one(std::integral_constant<int, Ws>{}), ...);
}
template <int C, int SPT>
__global__ void __launch_bounds__(1024) k_s2_count(const uint32_t* __restrict__ scalars, S2Geom g, uint32_t* __restrict__ cntA, uint32_t* __restrict__ bad) {
  constexpr int W = S2Shape<C>::W; constexpr int CH = 1024 * SPT;
  const int k = blockIdx.x, P = g.P;
  for (int j = threadIdx.x; j < W * P; j += 1024) s2_lds[j] = 0;
  __syncthreads();
  for (int i = 0; i < SPT; i++) {
    const size_t t = (size_t)k * CH + (size_t)i * 1024 + threadIdx.x;
    if (t >= g.n) break;
    const s2_u32x8 kk = s2_load_scalar(scalars + 8 * t);
    if (!s2_is_canonical(kk)) atomicAdd(bad, 1u);       // reported as SBN_EINVAL by the entry point (scalar.rs:87-95)
    s2_count_scalar<C>(kk, P, g.lo_log, std::make_integer_sequence<int, W>{});
  }
  __syncthreads();
  for (int j = threadIdx.x; j < W * P; j += 1024) cntA[(size_t)j * g.K + k] = s2_lds[j];
}

// exclusive scan of up to 1024 values, one per thread (threads beyond `count` pass 0); returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t s2_block_scan(uint32_t v, uint32_t* tmp /* 17 words of LDS */, uint32_t* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
  if (lane == 63) tmp[wv] = inc;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < nw; i++) { const uint32_t x = tmp[i]; tmp[i] = run; run += x; } tmp[16] = run; }
  __syncthreads();
  const uint32_t r = inc - v + tmp[wv];
  *total = tmp[16];
  __syncthreads();
  return r;
}
// pass B1: row j = (w, hi): exclusive prefix over the K chunks in place, total -> part_cnt[j]
__global__ void __launch_bounds__(256) k_s2_prefix_k(uint32_t* __restrict__ cntA, int K, uint32_t* __restrict__ part_cnt) {
  __shared__ uint32_t tmp[17];
  uint32_t* row = cntA + (size_t)blockIdx.x * K;
  const int per = (K + 255) / 256, b0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (int i = 0; i < per; i++) if (b0 + i < K) sum += row[b0 + i];
  uint32_t total; uint32_t run = s2_block_scan(sum, tmp, &total);
  for (int i = 0; i < per; i++) if (b0 + i < K) { const uint32_t v = row[b0 + i]; row[b0 + i] = run; run += v; }
  if (threadIdx.x == 0) part_cnt[blockIdx.x] = total;
}
// pass B2 (one block): per window, where each partition starts inside the window's n slots (part_off), and — over all windows —
// the index of each partition's first level-2 sub-chunk (sc_off[q], sc_off[W*P] = their number): a partition of cnt entries is
// cut into ceil(cnt / S2_SUB) sub-chunks, so level-2 blocks carry equal loads whatever the partition sizes are (the short top
// window puts all its entries into a few partitions; skewed scalars do the same anywhere).
constexpr uint32_t S2_SUB = 16384;
__global__ void __launch_bounds__(1024) k_s2_prefix_hi(const uint32_t* __restrict__ part_cnt, int W, int P, uint32_t* __restrict__ part_off, uint32_t* __restrict__ sc_off) {
  // ONE scan over the flat (window, partition) array instead of W scans one after the other (26 us on the critical path of a single
  // MSM): every thread owns a run of consecutive entries; part_off first receives the GLOBAL exclusive prefix (mod 2^32: only
  // differences are used) and then loses its window's start.
  __shared__ uint32_t tmp[17];
  __shared__ uint32_t wstart[32];
  const int n = W * P, per = (n + 1023) / 1024, i0 = (int)threadIdx.x * per, i1 = (i0 + per < n) ? i0 + per : n;
  uint32_t sv = 0, sk = 0;
  for (int i = i0; i < i1; i++) { const uint32_t v = part_cnt[i]; sv += v; sk += (v + S2_SUB - 1) / S2_SUB; }
  uint32_t tv, tk;
  uint32_t rv = s2_block_scan(sv, tmp, &tv), rk = s2_block_scan(sk, tmp, &tk);
  for (int i = i0; i < i1; i++) { const uint32_t v = part_cnt[i]; part_off[i] = rv; sc_off[i] = rk; rv += v; rk += (v + S2_SUB - 1) / S2_SUB; }
  if (threadIdx.x == 0) sc_off[(size_t)n] = tk;
  __syncthreads();                                   // the block's global stores are visible to the block
  if ((int)threadIdx.x < W) wstart[threadIdx.x] = part_off[(size_t)threadIdx.x * P];
  __syncthreads();
  for (int i = i0; i < i1; i++) part_off[i] -= wstart[i / P];
}

// level 1, pass C: the block's entries of window w, sorted by partition inside LDS and written out as runs.
// LDS: cnt[P] | base[P] | gdelta[P] | scan tmp[32] | stage_idx[CH] | stage_lo[CH] (u16) | stage_hi[CH] (u16)      (CH = 1024 * SPT)
__host__ __device__ inline size_t s2_scatter_lds_bytes(int P, int spt = S2_SPT) { return ((size_t)3 * P + 32) * 4 + (size_t)1024 * spt * 8; }
struct S2ScatterArgs {
  const uint32_t* cntA; const uint32_t* part_off; uint32_t* tmp_idx; uint16_t* tmp_lo;
  size_t n; int P, K, k, lo_log;
};
template <int C, int w, int SPT>
__device__ __forceinline__ void s2_scatter_window(const s2_u32x8 (&kk)[SPT], uint32_t (&carry)[SPT], const S2ScatterArgs& a) {
  constexpr int CH = 1024 * SPT;
  const int P = a.P, tid = threadIdx.x;
  uint32_t* cnt = s2_lds; uint32_t* base = cnt + P; uint32_t* gdelta = base + P; uint32_t* tmp = gdelta + P;
  uint32_t* stage_idx = tmp + 32;
  uint16_t* stage_lo = reinterpret_cast<uint16_t*>(stage_idx + CH); uint16_t* stage_hi = stage_lo + CH;
  for (int j = tid; j < P; j += 1024) cnt[j] = 0;
  __syncthreads();
  int d[SPT]; uint32_t rank[SPT];
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const int dd = s2_digit<C, w>(kk[i], carry[i]);
    d[i] = dd; rank[i] = 0;
    if (dd != 0) rank[i] = atomicAdd(&cnt[((dd < 0 ? -dd : dd) - 1) >> a.lo_log], 1u);
  }
  __syncthreads();
  uint32_t total;
  {
    const uint32_t v = tid < P ? cnt[tid] : 0;
    const uint32_t r = s2_block_scan(v, tmp, &total);
    if (tid < P) { base[tid] = r; gdelta[tid] = a.part_off[(size_t)w * P + tid] + a.cntA[((size_t)w * P + tid) * a.K + a.k] - r; }
  }
  __syncthreads();
  const size_t t0 = (size_t)a.k * CH + tid;
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    if (d[i] == 0) continue;
    const uint32_t b = (uint32_t)((d[i] < 0 ? -d[i] : d[i]) - 1), hi = b >> a.lo_log;
    const uint32_t pos = base[hi] + rank[i];
    stage_idx[pos] = (uint32_t)(t0 + (size_t)i * 1024) | (d[i] < 0 ? 0x80000000u : 0u);
    stage_lo[pos] = (uint16_t)(b & ((1u << a.lo_log) - 1u)); stage_hi[pos] = (uint16_t)hi;
  }
  __syncthreads();
  uint32_t* oi = a.tmp_idx + (size_t)w * a.n; uint16_t* ol = a.tmp_lo + (size_t)w * a.n;
  for (uint32_t j = tid; j < total; j += 1024) {
    const uint32_t pos = j + gdelta[stage_hi[j]];
    oi[pos] = stage_idx[j]; ol[pos] = stage_lo[j];
  }
  __syncthreads();
}
template <int C, int SPT, int... Ws>
__device__ __forceinline__ void s2_scatter_windows(const s2_u32x8 (&kk)[SPT], uint32_t (&carry)[SPT], const S2ScatterArgs& a, std::integer_sequence<int, Ws...>) {
  (s2_scatter_window<C, Ws, SPT>(kk, carry, a), ...);
}
template <int C, int SPT>
__global__ void __launch_bounds__(1024) k_s2_scatter(const uint32_t* __restrict__ scalars, S2Geom g, const uint32_t* __restrict__ cntA /* prefixed over k */,
                                                     const uint32_t* __restrict__ part_off, uint32_t* __restrict__ tmp_idx, uint16_t* __restrict__ tmp_lo) {
  S2ScatterArgs a; a.cntA = cntA; a.part_off = part_off; a.tmp_idx = tmp_idx; a.tmp_lo = tmp_lo; a.n = g.n; a.P = g.P; a.K = g.K; a.k = blockIdx.x; a.lo_log = g.lo_log;
  s2_u32x8 kk[SPT]; uint32_t carry[SPT];
#pragma unroll
  for (int i = 0; i < SPT; i++) {
    const size_t t = (size_t)blockIdx.x * (1024 * SPT) + (size_t)i * 1024 + threadIdx.x;
    carry[i] = 0;
    kk[i] = t < g.n ? s2_load_scalar(scalars + 8 * t) : (s2_u32x8)(0u);      // zero scalar: every digit 0, no entries
  }
  s2_scatter_windows<C, SPT>(kk, carry, a, std::make_integer_sequence<int, S2Shape<C>::W>{});
}

// level 2: a block = one sub-chunk (S2_SUB entries of one partition).  Consecutive workgroup ids go round-robin over the 8 XCDs,
// so the id is decoded as (xcd, slot) and groups of 16 consecutive sub-chunks — a partition's, mostly — run on one XCD, whose L2
// then sees all the stores into that slice of the output.
__device__ __forceinline__ bool s2_decode(const S2Geom& g, const uint32_t* __restrict__ sc_off, int& q, uint32_t& sidx, uint32_t& k2) {
  const uint32_t Lid = blockIdx.x, xcd = Lid & 7u, slot = Lid >> 3;
  sidx = (slot >> 4) * 128u + xcd * 16u + (slot & 15u);
  const int WP = g.W * g.P;
  if (sidx >= sc_off[WP]) return false;
  int lo = 0, hi = WP;                      // largest q with sc_off[q] <= sidx (empty partitions repeat an offset: take the last)
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sc_off[mid] <= sidx) lo = mid; else hi = mid; }
  q = lo; k2 = sidx - sc_off[q];
  return true;
}
__host__ __device__ inline uint32_t s2_level2_blocks(size_t max_subchunks) { return (uint32_t)(((max_subchunks + 127) / 128) * 128); }
__device__ __forceinline__ void s2_range(const S2Geom& g, const uint32_t* __restrict__ part_off, const uint32_t* __restrict__ part_cnt, int q, uint32_t k2, size_t& e0, size_t& e1) {
  const uint32_t start = part_off[q], cn = part_cnt[q];
  uint32_t a = k2 * S2_SUB, b = a + S2_SUB; if (a > cn) a = cn; if (b > cn) b = cn;
  const size_t w = (size_t)(q / g.P);
  e0 = w * g.n + start + a; e1 = w * g.n + start + b;
}
// blockhist[sidx * LO + lo]
__global__ void __launch_bounds__(1024) k_s2_hist(const uint16_t* __restrict__ tmp_lo, S2Geom g, const uint32_t* __restrict__ part_off, const uint32_t* __restrict__ part_cnt, const uint32_t* __restrict__ sc_off, uint32_t* __restrict__ blockhist) {
  __shared__ uint32_t h[S2_LO_MAX];
  int q; uint32_t sidx, k2; if (!s2_decode(g, sc_off, q, sidx, k2)) return;
  const int LO = 1 << g.lo_log;
  for (int j = threadIdx.x; j < LO; j += 1024) h[j] = 0;
  __syncthreads();
  size_t e0, e1; s2_range(g, part_off, part_cnt, q, k2, e0, e1);
  for (size_t e = e0 + threadIdx.x; e < e1; e += 1024) atomicAdd(&h[tmp_lo[e]], 1u);
  __syncthreads();
  uint32_t* o = blockhist + (size_t)sidx * LO;
  for (int j = threadIdx.x; j < LO; j += 1024) o[j] = h[j];
}
// one block per partition: exclusive prefix over its sub-chunks for every bucket (in place), the bucket sizes (hist) and the
// bucket start offsets inside the window (offs = partition start + exclusive prefix over the partition's buckets)
__global__ void __launch_bounds__(1024) k_s2_prefix2(uint32_t* __restrict__ blockhist, S2Geom g, const uint32_t* __restrict__ part_off, const uint32_t* __restrict__ sc_off, uint32_t* __restrict__ hist, uint32_t* __restrict__ offs) {
  __shared__ uint32_t tmp[17];
  const int q = blockIdx.x, LO = 1 << g.lo_log, w = q / g.P, hi = q - w * g.P;
  const uint32_t s0 = sc_off[q], nk = sc_off[q + 1] - s0;
  const int j0 = 2 * threadIdx.x;
  uint32_t r0 = 0, r1 = 0;
  if (j0 < LO) {
    uint2* col = reinterpret_cast<uint2*>(blockhist + (size_t)s0 * LO + j0);         // LO >= 2 and even: buckets j0, j0 + 1 are one 8-byte word
    for (uint32_t k = 0; k < nk; k++) { uint2* pq = col + (size_t)k * (LO / 2); const uint2 v = *pq; *pq = make_uint2(r0, r1); r0 += v.x; r1 += v.y; }
  }
  uint32_t total; const uint32_t r = s2_block_scan(r0 + r1, tmp, &total);
  if (j0 < LO) {
    const size_t b = ((size_t)w * g.P + hi) * LO + j0;
    hist[b] = r0; hist[b + 1] = r1;
    offs[b] = part_off[q] + r; offs[b + 1] = part_off[q] + r + r0;
  }
}
// Final placement, staged: the sub-chunk is taken in tiles of 1024 * EPT entries; a tile is sorted by lo inside LDS and leaves
// as runs (the entries of one bucket are consecutive in the stage AND at their destination), so a wave's 64 stores fall into a
// few contiguous pieces instead of 64 separate 4-byte L2 transactions (scattered dword stores run at ~5.6e10/s chip-wide, which
// is what bounded the unstaged form: 13.5 ms for the 8.7e8 entries of a 2^26 MSM).
// LDS: cnt[LO] | base[LO] | cur[LO] | scan tmp[32] | stage_idx[T] | stage_lo[T] (u16)
template <int EPT> __host__ __device__ inline size_t s2_place_lds_bytes(int lo_log) { return ((size_t)3 * ((size_t)1 << lo_log) + 32) * 4 + (size_t)1024 * EPT * 6; }
template <int EPT>
__global__ void __launch_bounds__(1024) k_s2_place(const uint16_t* __restrict__ tmp_lo, const uint32_t* __restrict__ tmp_idx, S2Geom g, const uint32_t* __restrict__ part_off, const uint32_t* __restrict__ part_cnt,
                                                   const uint32_t* __restrict__ sc_off, const uint32_t* __restrict__ blockhist /* prefixed over the partition's sub-chunks */,
                                                   const uint32_t* __restrict__ offs, uint32_t* __restrict__ sorted, size_t estride) {
  constexpr uint32_t T = 1024 * EPT;
  int q; uint32_t sidx, k2; if (!s2_decode(g, sc_off, q, sidx, k2)) return;
  const int LO = 1 << g.lo_log, tid = threadIdx.x;
  uint32_t* cnt = s2_lds; uint32_t* base = cnt + LO; uint32_t* cur = base + LO; uint32_t* tmp = cur + LO;
  uint32_t* stage_idx = tmp + 32; uint16_t* stage_lo = reinterpret_cast<uint16_t*>(stage_idx + T);
  const int w = q / g.P, hi = q - w * g.P;
  const size_t nb = (size_t)g.P << g.lo_log;
  const uint32_t* b0 = blockhist + (size_t)sidx * LO;
  const uint32_t* of = offs + (size_t)w * nb + ((size_t)hi << g.lo_log);
  for (int j = tid; j < LO; j += 1024) { cur[j] = of[j] + b0[j]; cnt[j] = 0; }
  __syncthreads();
  size_t e0, e1; s2_range(g, part_off, part_cnt, q, k2, e0, e1);
  uint32_t* out = sorted + (size_t)w * estride;
  for (size_t tile = e0; tile < e1; tile += T) {          // (uniform trip count per block)
    uint32_t lo[EPT], idx[EPT], rank[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      const size_t e = tile + (size_t)i * 1024 + tid;
      lo[i] = 0xffffffffu;
      if (e < e1) { lo[i] = tmp_lo[e]; idx[i] = tmp_idx[e]; rank[i] = atomicAdd(&cnt[lo[i]], 1u); }
    }
    __syncthreads();
    uint32_t total;
    {
      // LO <= 2048 counters, two per lane
      const int j0 = 2 * tid, j1 = 2 * tid + 1;
      const uint32_t v0 = j0 < LO ? cnt[j0] : 0, v1 = j1 < LO ? cnt[j1] : 0;
      const uint32_t r = s2_block_scan(v0 + v1, tmp, &total);
      if (j0 < LO) base[j0] = r;
      if (j1 < LO) base[j1] = r + v0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      if (lo[i] == 0xffffffffu) continue;
      const uint32_t pos = base[lo[i]] + rank[i];
      stage_idx[pos] = idx[i]; stage_lo[pos] = (uint16_t)lo[i];
    }
    __syncthreads();
    for (uint32_t j = tid; j < total; j += 1024) {
      const uint32_t l = stage_lo[j];
      out[cur[l] + (j - base[l])] = stage_idx[j];
    }
    __syncthreads();
    for (int j = tid; j < LO; j += 1024) { cur[j] += cnt[j]; cnt[j] = 0; }
    __syncthreads();
  }
}

}  // namespace sbn
