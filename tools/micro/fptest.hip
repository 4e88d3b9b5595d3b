// On-device check + throughput of sbn::fe_mul (asm even/odd accumulators) against a plain-C CIOS.
#include "../../spartan-bn254_amd/csrc/fp.cuh"
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;

template <class M> __device__ __forceinline__ Fe<M> ref_mul(const Fe<M>& a, const Fe<M>& b) {
  uint32_t p[8]; for (int i = 0; i < 8; i++) p[i] = modlimb<M>(i);
  uint32_t t[8] = {0,0,0,0,0,0,0,0};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t A = (uint64_t)a.v[0] * b.v[i] + t[0];
    uint32_t m = (uint32_t)A * M::NINV;
    uint64_t C = (uint64_t)m * p[0] + (uint32_t)A;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      A = (uint64_t)a.v[j] * b.v[i] + t[j] + (A >> 32);
      C = (uint64_t)m * p[j] + (uint32_t)A + (C >> 32);
      t[j - 1] = (uint32_t)C;
    }
    t[7] = (uint32_t)(C >> 32) + (uint32_t)(A >> 32);
  }
  Fe<M> r; fe_cond_sub_p<M>(r.v, t); return r;
}

template <class M> __global__ void k_check(const uint32_t* in, uint32_t* bad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  Fe<M> a = fe_load<M>(in + 16 * (size_t)i), b = fe_load<M>(in + 16 * (size_t)i + 8);
  Fe<M> x = fe_mul<M>(a, b), y = ref_mul<M>(a, b);
  if (!fe_eq<M>(fe_mul_deferred<M>(a, b), y)) atomicAdd(bad, 1);
  Fe<M> s = fe_add<M>(a, b), d = fe_sub<M>(s, b);   // (a+b)-b == a
  Fe<M> n0 = fe_add<M>(a, fe_neg<M>(a));
  Fe<M> rt = fe_from_mont<M>(fe_to_mont<M>(a));
  if (!fe_eq<M>(x, y) || !fe_eq<M>(d, a) || !fe_is_zero<M>(n0) || !fe_eq<M>(rt, a)) atomicAdd(bad, 1);
}
template <class M> __global__ void k_inv(const uint32_t* in, uint32_t* bad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  Fe<M> a = fe_load<M>(in + 16 * (size_t)i);
  if (fe_is_zero<M>(a)) return;
  Fe<M> ai = fe_inv<M>(a); Fe<M> one = fe_mul<M>(a, ai);
  if (!fe_eq<M>(one, fe_one<M>())) atomicAdd(bad, 1);
}
constexpr int MM_ITERS = 512;
template <class M, int VAR> __global__ void __launch_bounds__(256) k_rate(uint32_t* o, const uint32_t* in) {
  Fe<M> a = fe_load<M>(in + 16 * threadIdx.x), b = fe_load<M>(in + 16 * threadIdx.x + 8);
  for (int i = 0; i < MM_ITERS; i++) {
    if (VAR == 0) { a = fe_mul<M>(a, b); b = fe_mul<M>(b, a); }
    else if (VAR == 2) { a = fe_mul_deferred<M>(a, b); b = fe_mul_deferred<M>(b, a); }
    else { a = ref_mul<M>(a, b); b = ref_mul<M>(b, a); }
  }
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= a.v[j] ^ b.v[j];
  o[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class M> __global__ void __launch_bounds__(256) k_rate_add(uint32_t* o, const uint32_t* in) {
  Fe<M> a = fe_load<M>(in + 16 * threadIdx.x), b = fe_load<M>(in + 16 * threadIdx.x + 8);
  for (int i = 0; i < MM_ITERS * 4; i++) { a = fe_add<M>(a, b); b = fe_sub<M>(b, a); }
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= a.v[j] ^ b.v[j];
  o[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static const uint64_t PQ[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t PR[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static uint64_t rs = 88172645463325252ull;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static bool lt(const uint64_t a[4], const uint64_t p[4]) { for (int i = 3; i >= 0; i--) { if (a[i] < p[i]) return true; if (a[i] > p[i]) return false; } return false; }
static void gen(std::vector<uint32_t>& h, int n, const uint64_t p[4]) {
  h.resize((size_t)n * 16);
  for (int i = 0; i < 2 * n; i++) {
    uint64_t v[4];
    int mode = i % 16;
    do {
      for (int j = 0; j < 4; j++) v[j] = rnd();
      v[3] &= 0x3fffffffffffffffull;
      if (mode == 1) { for (int j = 0; j < 4; j++) v[j] = p[j]; v[0] -= 1 + (rnd() % 3); }         // p-1..p-3
      if (mode == 2) { v[0] = rnd() % 4; v[1] = v[2] = v[3] = 0; }                                  // tiny
      if (mode == 3) { for (int j = 0; j < 4; j++) v[j] = 0xffffffffffffffffull; v[3] = p[3] - 1; } // dense ones
      if (mode == 4) { v[0] = v[1] = v[2] = 0xffffffffffffffffull; v[3] = p[3] - 1 - (rnd() & 0xff); }
    } while (!lt(v, p));
    for (int j = 0; j < 4; j++) { h[(size_t)i * 8 + 2 * j] = (uint32_t)v[j]; h[(size_t)i * 8 + 2 * j + 1] = (uint32_t)(v[j] >> 32); }
  }
}
template <class K> void rate(const char* name, K kern, int blocks, double ops, uint32_t* d_o, uint32_t* d_in) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_o, d_in); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_o, d_in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  double tot = ops * blocks * 256.0;
  printf("%-17s blocks=%5d %.3f ms %.3e ops/s  %.0f cyc/op/wave@2.4GHz\n", name, blocks, ms, tot / (ms * 1e-3), (ms * 1e-3) * 2.4e9 * 1024 / (tot / 64));
}
int main() {
  const int n = 1 << 20;
  uint32_t *d_in, *d_bad, *d_o; CK(hipMalloc(&d_in, (size_t)n * 64)); CK(hipMalloc(&d_bad, 4)); CK(hipMalloc(&d_o, 64 << 20));
  std::vector<uint32_t> h; int fails = 0;
  for (int f = 0; f < 2; f++) {
    gen(h, n, f ? PR : PQ);
    CK(hipMemcpy(d_in, h.data(), (size_t)n * 64, hipMemcpyHostToDevice)); CK(hipMemset(d_bad, 0, 4));
    if (f) hipLaunchKernelGGL(k_check<FrP>, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad, n);
    else hipLaunchKernelGGL(k_check<FqP>, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad, n);
    uint32_t bad; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("%s mul/add/sub/mont check: %u mismatches of %d\n", f ? "Fr" : "Fq", bad, n); fails += bad;
    CK(hipMemset(d_bad, 0, 4));
    if (f) hipLaunchKernelGGL(k_inv<FrP>, dim3(4096 / 256), dim3(256), 0, 0, d_in, d_bad, 4096);
    else hipLaunchKernelGGL(k_inv<FqP>, dim3(4096 / 256), dim3(256), 0, 0, d_in, d_bad, 4096);
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("%s inv check: %u mismatches of 4096\n", f ? "Fr" : "Fq", bad); fails += bad;
  }
  for (int occ : {1, 2, 3, 4, 8}) {      // occ = waves per SIMD (256-thread blocks per CU)
    rate("fe_mul(asm)", k_rate<FqP, 0>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("fe_mul(deferred)", k_rate<FqP, 2>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    if (occ == 1 || occ == 8) rate("fe_mul(C cios)", k_rate<FqP, 1>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("fe_add+sub", k_rate_add<FqP>, 256 * occ, 8.0 * MM_ITERS, d_o, d_in);
  }
  printf(fails ? "FPTEST FAIL\n" : "FPTEST OK\n");
  return fails != 0;
}
