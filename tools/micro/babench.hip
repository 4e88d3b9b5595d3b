// Batched-affine summation against the XYZZ mixed addition, measured, on the access pattern of the lookup
// commit: every operand is a random 64-byte affine point of a table far larger than the caches.
//   k_madd    what k_comb_rows does: one accumulator per lane, acc += table[idx] (8M + 2S, no inversion), next point prefetched
//   k_ba<B>   first tree level of a batched-affine sum: each lane adds B PAIRS of table points with ONE inversion (Montgomery's
//             trick): forward pass dx_i = x2 - x1, running product stored to a global scratch (coalesced, 32 B per pair);
//             one Fermat inversion per lane; backward pass re-reads the two points and the running product and finishes the
//             affine addition (5M + 1S per pair + the inversion's share, ~380/B products), result stored (64 B, coalesced).
//             Keeping the operands instead of re-reading them would need B x 128 B per lane of on-chip memory: with 512 lanes
//             per CU that is 4 MB per CU at B = 64 — there is 160 KB of LDS and 512 KB of registers.
// Points are random field elements (not on the curve): timing only.  Reported: additions per second, whole chip.
#include "../../spartan-bn254_amd/csrc/g1.cuh"
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;

__device__ __forceinline__ uint32_t hash32(uint32_t a, uint32_t b) {
  uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
  x ^= x >> 15; x *= 0xC2B2AE35u; x ^= x >> 13;
  return x;
}
constexpr int MADD_N = 256;
__global__ void __launch_bounds__(256) k_madd(const uint32_t* __restrict__ table, uint32_t mask, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ acc = xyzz_inf();
  Affine nxt = aff_load(table + 16 * (size_t)(hash32(tid, 0) & mask));
  for (int i = 0; i < MADD_N; i++) {
    const Affine p = nxt;
    nxt = aff_load(table + 16 * (size_t)(hash32(tid, i + 1) & mask));
    xyzz_madd(acc, p, (i & 7) == 3);
  }
  xyzz_store(out + 32 * (size_t)tid, acc);
}
template <int B> __global__ void __launch_bounds__(256) k_ba(const uint32_t* __restrict__ table, uint32_t mask, uint32_t* __restrict__ pref, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t T = (size_t)gridDim.x * blockDim.x;
  Fq run = fe_one<FqP>();
  for (int i = 0; i < B; i++) {
    const Fq x1 = fe_load<FqP>(table + 16 * (size_t)(hash32(tid, 2 * i) & mask)), x2 = fe_load<FqP>(table + 16 * (size_t)(hash32(tid, 2 * i + 1) & mask));
    run = fe_mulu(run, fe_normu(fe_subb<FqP, 2, 1>(x2, x1)));
    fe_store_packed<FqP>(pref + 8 * ((size_t)i * T + tid), run);
  }
  Fq inv = fe_inv(run);
  for (int i = B - 1; i >= 0; i--) {
    const Affine p = aff_load(table + 16 * (size_t)(hash32(tid, 2 * i) & mask)), q = aff_load(table + 16 * (size_t)(hash32(tid, 2 * i + 1) & mask));
    const Fq dx = fe_normu(fe_subb<FqP, 2, 1>(q.x, p.x));
    const Fq prev = i ? fe_load<FqP>(pref + 8 * ((size_t)(i - 1) * T + tid)) : fe_one<FqP>();
    const Fq di = fe_mulu(inv, prev);                 // 1 / dx_i
    inv = fe_mulu(inv, dx);
    const Fq lam = fe_mulu(fe_normu(fe_subb<FqP, 2, 1>(q.y, p.y)), di);
    const Fq x3 = fe_normu(fe_subb<FqP, 4, 2>(fe_squ(lam), fe_add_lazy(p.x, q.x)));
    const Fq y3 = fe_normu(fe_subb<FqP, 2, 1>(fe_mulu(lam, fe_normu(fe_subb<FqP, 6, 1>(p.x, x3))), p.y));
    fe_store_packed<FqP>(out + 16 * ((size_t)i * T + tid), x3); fe_store_packed<FqP>(out + 16 * ((size_t)i * T + tid) + 8, y3);
  }
}
__global__ void k_fill(uint32_t* t, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t v = hash32((uint32_t)i, (uint32_t)(i >> 32) + 17u);
    if ((i & 7) == 7) v &= 0x0fffffffu;              // every element below 2^252 < p
    t[i] = v;
  }
}
template <class F> float timed(F launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
template <int B> void run_ba(const uint32_t* table, uint32_t mask, uint32_t* pref, uint32_t* out, int blocks) {
  const float ms = timed([&] { hipLaunchKernelGGL(k_ba<B>, dim3(blocks), dim3(256), 0, 0, table, mask, pref, out); }, 3);
  const double adds = (double)blocks * 256 * B;
  printf("batched affine  B=%3d  blocks/CU=%d  %.3f ms  %.3e additions/s  (gathers %.3e /s)\n", B, blocks / 256, ms, adds / (ms * 1e-3), 4 * adds / (ms * 1e-3));
}
int main() {
  const int logn = 26; const size_t npts = (size_t)1 << logn; const uint32_t mask = (uint32_t)(npts - 1);
  uint32_t *table, *pref, *out;
  CK(hipMalloc(&table, npts * 64));
  const int maxblocks = 256 * 8; const size_t T = (size_t)maxblocks * 256;
  CK(hipMalloc(&pref, T * 256 * 32)); CK(hipMalloc(&out, T * 256 * 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, npts * 16); CK(hipDeviceSynchronize());
  for (int occ : {2, 4, 8}) {
    const int blocks = 256 * occ;
    const float ms = timed([&] { hipLaunchKernelGGL(k_madd, dim3(blocks), dim3(256), 0, 0, (const uint32_t*)table, mask, out); }, 3);
    const double adds = (double)blocks * 256 * MADD_N;
    printf("xyzz mixed add         blocks/CU=%d  %.3f ms  %.3e additions/s  (gathers %.3e /s)\n", occ, ms, adds / (ms * 1e-3), adds / (ms * 1e-3));
  }
  for (int occ : {2, 4}) {
    run_ba<32>(table, mask, pref, out, 256 * occ);
    run_ba<64>(table, mask, pref, out, 256 * occ);
    run_ba<128>(table, mask, pref, out, 256 * occ);
    run_ba<256>(table, mask, pref, out, 256 * occ);
  }
  return 0;
}
