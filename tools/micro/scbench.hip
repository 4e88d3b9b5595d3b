// Where the combined sumcheck round (k_sc_comb_bind_eval) spends its time: variants of its loop on the keyless shape (12 "par"
// instances, tables of 2^21, q = 2^19 index pairs), each timed alone.  Timing only (the variants that skip work give wrong sums).
//   base     the shipped loop (1-deep register prefetch)
//   nomac    without the three column accumulations   -> what the evaluation costs
//   nostore  without the stores of the bound values   -> what the write stream costs
//   nobind   loads + accumulations only               -> the memory side alone
//   depth2   DD kept in LDS between instances, two tables in flight per thread
//   stagger  waves of odd blocks start half a table step late
#include "../../spartan-bn254_amd/csrc/sumcheck_comb_kernels.cuh"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;
enum { NO_MAC = 1, NO_STORE = 2, NO_BIND = 4, DEPTH2 = 8, STAGGER = 16, ACC_REGS = 32 };

struct ColsLds { uint32_t w[2 * NC][256]; };
__device__ __forceinline__ void cols_to_lds(ColsLds& L, const Cols& s) {
#pragma unroll
  for (int k = 0; k < NC; k++) { L.w[2 * k][threadIdx.x] = (uint32_t)s.c[k]; L.w[2 * k + 1][threadIdx.x] = (uint32_t)(s.c[k] >> 32); }
}
__device__ __forceinline__ void cols_from_lds(const ColsLds& L, Cols& s) {
#pragma unroll
  for (int k = 0; k < NC; k++) s.c[k] = (uint64_t)L.w[2 * k][threadIdx.x] | ((uint64_t)L.w[2 * k + 1][threadIdx.x] << 32);
}

template <int F>
__global__ void __launch_bounds__(256, 2) k_var(const ScCombGroup* __restrict__ groups, size_t q, ScScalar rmont, uint32_t* __restrict__ sink) {
  const ScCombGroup* __restrict__ g = groups + blockIdx.y;
  const Fr r = fr_from_words(rmont);
  const uint32_t n = g->n;
  __shared__ ScLdsAcc acc;
  __shared__ ColsLds ddl;
  Fr e0 = fe_zero<FrP>(), e2 = e0, e3 = e0;
  uint32_t c0 = 0, c2 = 0, c3 = 0;
  if (!(F & ACC_REGS)) sc_lds_acc_zero(acc);
  uint32_t cnt = 0;
  if ((F & STAGGER) && (blockIdx.x & 1)) __builtin_amdgcn_s_sleep(127);
#define SC_PIN() __builtin_amdgcn_sched_barrier(0)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
    Cols LL, HH, DD; cols_zero(LL); cols_zero(HH); cols_zero(DD);
    if (F & DEPTH2) cols_to_lds(ddl, DD);
    ScQuad zn = sc_quad_load<false>(g->a_src[0], i, q);
    ScQuad zn2;
    if (F & DEPTH2) zn2 = sc_quad_load<false>(g->b_src[0], i, q);
    uint32_t pend = 0;
    for (uint32_t k = 0; k < n; k++) {
      ScPair a, b;
      const uint32_t kn = k + 1 < n ? k + 1 : k;
      {
        const ScQuad z = zn;
        if (F & DEPTH2) { zn = zn2; zn2 = sc_quad_load<false>(g->a_src[kn], i, q); } else zn = sc_quad_load<false>(g->b_src[k], i, q);
        SC_PIN();
        if (F & NO_BIND) { a.lo = z.z0; a.hi = z.z1; } else { a.lo = sc_bind1(z.z0, z.z2, r); a.hi = sc_bind1(z.z1, z.z3, r); }
        uint32_t* d = g->a_dst[k];
        if (!(F & NO_STORE)) { fe_gstore_packed<FrP>(d + 8 * i, a.lo); fe_gstore_packed<FrP>(d + 8 * (i + q), a.hi); }
      }
      {
        const ScQuad z = zn;
        if (F & DEPTH2) { zn = zn2; zn2 = sc_quad_load<false>(g->b_src[kn], i, q); } else zn = sc_quad_load<false>(g->a_src[kn], i, q);
        SC_PIN();
        if (F & NO_BIND) { b.lo = z.z0; b.hi = z.z1; } else { b.lo = sc_bind1(z.z0, z.z2, r); b.hi = sc_bind1(z.z1, z.z3, r); }
        uint32_t* d = g->b_dst[k];
        if (!(F & NO_STORE)) { fe_gstore_packed<FrP>(d + 8 * i, b.lo); fe_gstore_packed<FrP>(d + 8 * (i + q), b.hi); }
      }
      if (!(F & NO_MAC)) {
        if (pend == 6) { cols_carry(LL); cols_carry(HH); if (F & DEPTH2) { cols_from_lds(ddl, DD); cols_carry(DD); cols_to_lds(ddl, DD); } else cols_carry(DD); pend = 0; }
        pend++;
        const Fr da = fe_normu(fe_subb<FrP, 3, 1>(a.hi, a.lo)), db = fe_normu(fe_subb<FrP, 3, 1>(b.hi, b.lo));
        cols_mac<FrP>(LL, a.lo, b.lo); cols_mac<FrP>(HH, a.hi, b.hi);
        if (F & DEPTH2) { cols_from_lds(ddl, DD); cols_mac<FrP>(DD, da, db); cols_to_lds(ddl, DD); } else cols_mac<FrP>(DD, da, db);
      } else { LL.c[0] += a.lo.v[0] + b.hi.v[1]; HH.c[0] += a.hi.v[2] + b.lo.v[3]; }
    }
    if (F & DEPTH2) cols_from_lds(ddl, DD);
    const Fr cl = fe_gload<FrP>(g->c + 8 * i), ch = fe_gload<FrP>(g->c + 8 * (i + q));
    const ScS S = sc_bracket_points(cols_reduce<FrP>(LL), cols_reduce<FrP>(HH), cols_reduce<FrP>(DD));
    const ScPts pc = sc_points_u(cl, ch);
    if (F & ACC_REGS) { fr_acc32(e0, fe_mulu(S.s0, cl), c0); fr_acc32(e2, fe_mulu(S.s2, pc.v2), c2); fr_acc32(e3, fe_mulu(S.s3, pc.v3), c3); }
    else { sc_lds_acc_add(acc, 0, fe_mulu(S.s0, cl), cnt); sc_lds_acc_add(acc, 1, fe_mulu(S.s2, pc.v2), cnt); sc_lds_acc_add(acc, 2, fe_mulu(S.s3, pc.v3), cnt); }
    cnt++;
  }
#undef SC_PIN
  Fr s = (F & ACC_REGS) ? fe_add(fe_add(e0, e2), e3) : fe_add(fe_add(sc_lds_acc_get(acc, 0), sc_lds_acc_get(acc, 1)), sc_lds_acc_get(acc, 2));
  if (s.v[0] == 0x12345678u && s.v[3] == 77u) sink[threadIdx.x] = s.v[1];          // keeps the sums alive
}

// memory pattern alone: per table 4 element reads (i, i+q, i+2q, i+3q) and 2 element writes (i, i+q), all bytes used, no field arithmetic.
// LAYOUT 0: 32-byte elements, lane i owns element i (two 16-byte loads per element, each touching half of every cache line);
// LAYOUT 1: "split" blocks of 64 elements: their 64 low halves (1 KiB) then their 64 high halves (1 KiB): every wave-instruction contiguous.
template <int LAYOUT> __device__ __forceinline__ const sbn_g_u32x4* el_ptr(const uint32_t* base, size_t idx, int half) {
  if (LAYOUT == 0) return (const sbn_g_u32x4*)(base + 8 * idx) + half;
  return (const sbn_g_u32x4*)(base + 8 * (idx & ~(size_t)63)) + 64 * half + (idx & 63);
}
template <int LAYOUT>
__global__ void __launch_bounds__(256, 2) k_pattern(const ScCombGroup* __restrict__ groups, size_t q, uint32_t* __restrict__ sink) {
  const ScCombGroup* __restrict__ g = groups + blockIdx.y;
  const uint32_t n = g->n;
  sbn_u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
    for (uint32_t k = 0; k < 2 * n; k++) {
      const uint32_t* src = k < n ? g->a_src[k] : g->b_src[k - n];
      uint32_t* dst = k < n ? g->a_dst[k] : g->b_dst[k - n];
      sbn_u32x4 v[8];
#pragma unroll
      for (int e = 0; e < 4; e++) { v[2 * e] = *el_ptr<LAYOUT>(src, i + e * q, 0); v[2 * e + 1] = *el_ptr<LAYOUT>(src, i + e * q, 1); }
      const sbn_u32x4 lo0 = v[0] ^ v[4], lo1 = v[1] ^ v[5], hi0 = v[2] ^ v[6], hi1 = v[3] ^ v[7];
      *(sbn_g_u32x4*)el_ptr<LAYOUT>(dst, i, 0) = lo0; *(sbn_g_u32x4*)el_ptr<LAYOUT>(dst, i, 1) = lo1;
      *(sbn_g_u32x4*)el_ptr<LAYOUT>(dst, i + q, 0) = hi0; *(sbn_g_u32x4*)el_ptr<LAYOUT>(dst, i + q, 1) = hi1;
      acc ^= lo0 ^ hi1;
    }
  }
  if (acc.x == 0x12345678u && acc.y == 77u) sink[threadIdx.x] = acc.z;
}
// LAYOUT 2 "tile-interleaved": a table of 4 q entries stores element i + e q (i < q, e < 4) at (i / 64) * 256 + e * 64 + i % 64: the four
// elements an index needs sit in ONE contiguous 8-KiB group per wavefront (one read stream per table instead of four), and the bound
// table (2 q = 4 q' entries) is written in the same layout for q' = q / 2: index i = j + b q' writes tiles b and b + 2 of group j / 64.
__global__ void __launch_bounds__(256, 2) k_pattern_tile(const ScCombGroup* __restrict__ groups, size_t q, uint32_t* __restrict__ sink) {
  const ScCombGroup* __restrict__ g = groups + blockIdx.y;
  const uint32_t n = g->n;
  const size_t qh = q / 2;
  sbn_u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
    const size_t rbase = (i >> 6) * 256 + (i & 63);
    const size_t b = i >= qh ? 1 : 0, j = i - b * qh;
    const size_t wbase = (j >> 6) * 256 + b * 64 + (j & 63);
    for (uint32_t k = 0; k < 2 * n; k++) {
      const uint32_t* src = k < n ? g->a_src[k] : g->b_src[k - n];
      uint32_t* dst = k < n ? g->a_dst[k] : g->b_dst[k - n];
      sbn_u32x4 v[8];
#pragma unroll
      for (int e = 0; e < 4; e++) { v[2 * e] = *el_ptr<0>(src, rbase + 64 * e, 0); v[2 * e + 1] = *el_ptr<0>(src, rbase + 64 * e, 1); }
      const sbn_u32x4 lo0 = v[0] ^ v[4], lo1 = v[1] ^ v[5], hi0 = v[2] ^ v[6], hi1 = v[3] ^ v[7];
      *(sbn_g_u32x4*)el_ptr<0>(dst, wbase, 0) = lo0; *(sbn_g_u32x4*)el_ptr<0>(dst, wbase, 1) = lo1;
      *(sbn_g_u32x4*)el_ptr<0>(dst, wbase + 128, 0) = hi0; *(sbn_g_u32x4*)el_ptr<0>(dst, wbase + 128, 1) = hi1;
      acc ^= lo0 ^ hi1;
    }
  }
  if (acc.x == 0x12345678u && acc.y == 77u) sink[threadIdx.x] = acc.z;
}
static void run_pattern_tile(const char* name, const ScCombGroup* dg, size_t q, unsigned gx, uint32_t* sink, double bytes) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  static void* junk = nullptr; if (!junk) CK(hipMalloc(&junk, (size_t)6 << 30));
  float sum = 0;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipMemsetAsync(junk, rep, (size_t)6 << 30, 0));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pattern_tile, dim3(gx, 1), dim3(256), 0, 0, dg, q, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) sum += ms;
  }
  printf("%-28s q=2^%d gx=%u: COLD %8.1f us  %6.2f TB/s\n", name, (int)(31 - __builtin_clz((unsigned)q)), gx, sum / 3 * 1e3, bytes / (sum / 3 * 1e-3) / 1e12);
}
template <int LAYOUT> static void run_pattern(const char* name, const ScCombGroup* dg, size_t q, unsigned gx, uint32_t* sink, double bytes) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  static void* junk = nullptr; if (!junk) CK(hipMalloc(&junk, (size_t)6 << 30));
  float sum = 0;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipMemsetAsync(junk, rep, (size_t)6 << 30, 0));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pattern<LAYOUT>, dim3(gx, 1), dim3(256), 0, 0, dg, q, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) sum += ms;
  }
  printf("%-28s q=2^%d gx=%u: COLD %8.1f us  %6.2f TB/s\n", name, (int)(31 - __builtin_clz((unsigned)q)), gx, sum / 3 * 1e3, bytes / (sum / 3 * 1e-3) / 1e12);
}

template <int F> static float run(const char* name, const ScCombGroup* dg, int G, size_t q, unsigned gx, uint32_t* sink, double bytes) {
  ScScalar rs; for (int k = 0; k < 8; k++) rs.v[k] = 0x1234567u * (k + 3); rs.v[7] &= 0x0fffffff;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_var<F>, dim3(gx, G), dim3(256), 0, 0, dg, q, rs, sink);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  if (getenv("SCB_COLD")) {
    // every timed launch behind a 6 GiB memset of another buffer: TLB and caches hold nothing of the tables (as in a prover, where
    // every round touches buffers the previous launch did not)
    static void* junk = nullptr; if (!junk) CK(hipMalloc(&junk, (size_t)6 << 30));
    float sum = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemsetAsync(junk, rep, (size_t)6 << 30, 0));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_var<F>, dim3(gx, G), dim3(256), 0, 0, dg, q, rs, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms;
    }
    printf("%-28s q=2^%d groups=%d gx=%u: COLD %8.1f us  %6.2f TB/s (algorithmic)\n", name, (int)(31 - __builtin_clz((unsigned)q)), G, gx, sum / 3 * 1e3, bytes / (sum / 3 * 1e-3) / 1e12);
  }
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_var<F>, dim3(gx, G), dim3(256), 0, 0, dg, q, rs, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-28s q=2^%d groups=%d gx=%u: %8.1f us  %6.2f TB/s (algorithmic)\n", name, (int)(31 - __builtin_clz((unsigned)q)), G, gx, best * 1e3, bytes / (best * 1e-3) / 1e12);
  return best;
}

int main(int argc, char** argv) {
  const int logn = argc > 1 ? atoi(argv[1]) : 21, NI = 12;
  const size_t n = (size_t)1 << logn, q = n / 4;
  std::vector<uint32_t*> src(2 * NI), dst(2 * NI);
  std::vector<uint32_t> h(n * 8); uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
  for (size_t i = 0; i < n; i++) h[8 * i + 7] &= 0x1fffffff;
  for (int t = 0; t < 2 * NI; t++) { CK(hipMalloc(&src[t], n * 32)); CK(hipMalloc(&dst[t], n * 16)); CK(hipMemcpy(src[t], h.data(), n * 32, hipMemcpyHostToDevice)); }
  uint32_t* cbuf; CK(hipMalloc(&cbuf, n * 16)); CK(hipMemcpy(cbuf, h.data(), n * 16, hipMemcpyHostToDevice));
  uint32_t* sink; CK(hipMalloc(&sink, 4096));
  const double bytes = (double)2 * NI * n * 32 * 1.5 + (double)n * 16;
  for (int G : {1, 2, 4}) {
    std::vector<ScCombGroup> hg(G); memset(hg.data(), 0, G * sizeof(ScCombGroup));
    const int per = NI / G;
    for (int g = 0; g < G; g++) {
      hg[g].n = per; hg[g].c = cbuf;
      for (int k = 0; k < per; k++) { const int i = g * per + k; hg[g].a_src[k] = src[i]; hg[g].b_src[k] = src[NI + i]; hg[g].a_dst[k] = dst[i]; hg[g].b_dst[k] = dst[NI + i]; }
    }
    ScCombGroup* dg; CK(hipMalloc(&dg, G * sizeof(ScCombGroup))); CK(hipMemcpy(dg, hg.data(), G * sizeof(ScCombGroup), hipMemcpyHostToDevice));
    const unsigned gx = (unsigned)((q + 255) / 256 < 2048u / G ? (q + 255) / 256 : 2048u / G);
    if (G == 1) {
      for (unsigned g2 : {512u, 1024u, 2048u}) if (g2 <= (q + 255) / 256) {
        run_pattern<0>("pattern: 32-B elements", dg, q, g2, sink, bytes - (double)n * 16);
        run_pattern<1>("pattern: split halves", dg, q, g2, sink, bytes - (double)n * 16);
        run_pattern_tile("pattern: tile-interleaved", dg, q, g2, sink, bytes - (double)n * 16);
      }
      run<0>("base", dg, G, q, gx, sink, bytes);
      run<ACC_REGS>("base, sums in registers", dg, G, q, gx, sink, bytes);
      run<NO_MAC>("nomac", dg, G, q, gx, sink, bytes);
      run<NO_STORE>("nostore", dg, G, q, gx, sink, bytes);
      run<NO_BIND>("nobind", dg, G, q, gx, sink, bytes);
      run<NO_BIND | NO_MAC>("nobind nomac (copy)", dg, G, q, gx, sink, bytes);
      run<DEPTH2>("depth2", dg, G, q, gx, sink, bytes);
      run<STAGGER>("stagger", dg, G, q, gx, sink, bytes);
      run<DEPTH2 | STAGGER>("depth2 stagger", dg, G, q, gx, sink, bytes);
      for (unsigned g2 : {512u, 1024u, 4096u}) if (g2 <= (q + 255) / 256) { run<0>("base", dg, G, q, g2, sink, bytes); run<DEPTH2>("depth2", dg, G, q, g2, sink, bytes); }
    } else { run<0>("base", dg, G, q, gx, sink, bytes); run<DEPTH2>("depth2", dg, G, q, gx, sink, bytes); }
  }
  return 0;
}
