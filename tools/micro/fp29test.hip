// Prototype + throughput of a 9 x 29-bit-limb Montgomery product (R = 2^261) against the shipped 8 x 32-bit sbn::fe_mul.
// Why: tools/micro/ibench.hip shows that on gfx950 every VOP3 / carry-writing VALU instruction (v_mad_u64_u32, v_addc_co_u32,
// v_mul_lo_u32, 64-bit shifts/adds) issues at ~4.2 cycles per wave-instruction and only plain VOP2 ops (v_add_u32, v_and_b32) at
// ~2.3: a carry capture costs as much as a multiply.  29-bit limbs leave 6 spare bits in a 64-bit column: 18 products per column
// never overflow, so the 105 captures of the 32-bit-limb product disappear (at the price of 162 instead of 128 mads) and
// additions become 9 plain v_add_u32 with no carry chain.
#include "legacy/fp32.cuh"      // the round-1 layer this prototype was measured against (the shipped fp.cuh IS the 29-bit layer now)
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn32;

constexpr uint32_t MASK29 = (1u << 29) - 1;
struct F29 { uint32_t v[9]; };

// limbs of p in radix 2^29 and -p^-1 mod 2^29, from the 32-bit limbs (compile-time helpers)
template <class M> __device__ __forceinline__ constexpr uint32_t p29(int k) {
  // bits [29k, 29k+29) of the modulus
  const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
  uint64_t x = (w < 8 ? (uint64_t)modlimb<M>(w) : 0) | (w + 1 < 8 ? (uint64_t)modlimb<M>(w + 1) << 32 : 0);
  return (uint32_t)(x >> sh) & MASK29;
}

template <class M> __device__ __forceinline__ F29 unpack29(const Fe<M>& a) {
  F29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
    uint64_t x = (w < 8 ? (uint64_t)a.v[w] : 0) | (w + 1 < 8 ? (uint64_t)a.v[w + 1] << 32 : 0);
    r.v[k] = (uint32_t)(x >> sh) & MASK29;
  }
  return r;
}
// limb-normalise (carry propagate) and pack a value < 2^256 into 8 x 32
template <class M> __device__ __forceinline__ Fe<M> pack29(F29 a) {
#pragma unroll
  for (int k = 0; k < 8; k++) { a.v[k + 1] += a.v[k] >> 29; a.v[k] &= MASK29; }
  Fe<M> r;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    // bits [32w, 32w+32): from limbs
    const int lo = (32 * w) / 29, sh = 32 * w - 29 * lo;
    uint64_t x = (uint64_t)a.v[lo] >> sh;
    if (lo + 1 < 9) x |= (uint64_t)a.v[lo + 1] << (29 - sh);
    if (lo + 2 < 9 && 58 - sh < 32) x |= (uint64_t)a.v[lo + 2] << (58 - sh);
    r.v[w] = (uint32_t)x;
  }
  return r;
}

template <class M, bool SQR> __device__ __forceinline__ F29 mul29(const F29& a, const F29& b) {
  constexpr uint32_t NINV29 = M::NINV & MASK29;       // -p^-1 mod 2^29 is the low 29 bits of -p^-1 mod 2^32
  uint64_t c[9];
  uint32_t a2[9];
  if (SQR) {
#pragma unroll
    for (int k = 0; k < 9; k++) a2[k] = a.v[k] << 1;
  }
#pragma unroll
  for (int i = 0; i < 9; i++) {
    if (!SQR) {
#pragma unroll
      for (int k = 0; k < 9; k++) {
        if (i == 0 || k == 8) c[k] = (uint64_t)a.v[k] * b.v[i];
        else c[k] += (uint64_t)a.v[k] * b.v[i];
      }
    } else {
      // squaring by rows: row i adds a_i^2 at window position i... written in the same sliding-window form:
      // window position k of round i is column i + k; products a_j a_l with j + l = i + k, j = i fixed as the "row": use the
      // symmetric form: row i contributes a_i * a_i at k = i? (kept simple here: doubled cross terms a2[k]*a_i for k > i, a_i^2 for k = i, nothing for k < i)
#pragma unroll
      for (int k = 0; k < 9; k++) {
        uint64_t t = 0; bool has = false;
        if (k == i) { t = (uint64_t)a.v[k] * a.v[i]; has = true; }
        else if (k > i) { t = (uint64_t)a2[k] * a.v[i]; has = true; }
        if (i == 0 || k == 8) c[k] = has ? t : 0; else if (has) c[k] += t;
      }
    }
    const uint32_t m = ((uint32_t)c[0] * NINV29) & MASK29;
#pragma unroll
    for (int k = 0; k < 9; k++) c[k] += (uint64_t)m * p29<M>(k);
    c[1] += c[0] >> 29;
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = c[k + 1];
  }
  F29 r;
#pragma unroll
  for (int k = 0; k < 8; k++) { r.v[k] = (uint32_t)c[k] & MASK29; if (k < 7) c[k + 1] += c[k] >> 29; else r.v[8] = (uint32_t)(c[7] >> 29); }
  return r;
}
template <class M> __device__ __forceinline__ F29 add29(const F29& a, const F29& b) { F29 r; for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + b.v[k]; return r; }

template <class M> __global__ void k_check(const uint32_t* in, uint32_t* bad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  Fe<M> a = fe_load<M>(in + 16 * (size_t)i), b = fe_load<M>(in + 16 * (size_t)i + 8);
  Fe<M> want = fe_mul<M>(a, b);                                    // a b 2^-256
  F29 r = mul29<M, false>(unpack29<M>(a), unpack29<M>(b));          // a b 2^-261, < 2p
  Fe<M> got = pack29<M>(r);
  { uint32_t t[8]; for (int k = 0; k < 8; k++) t[k] = got.v[k]; fe_cond_sub_p<M>(got.v, t); }
  for (int k = 0; k < 5; k++) got = fe_dbl<M>(got);                 // * 2^5
  Fe<M> wsq = fe_mul<M>(a, a);
  Fe<M> gsq = pack29<M>(mul29<M, true>(unpack29<M>(a), unpack29<M>(a)));
  { uint32_t t[8]; for (int k = 0; k < 8; k++) t[k] = gsq.v[k]; fe_cond_sub_p<M>(gsq.v, t); }
  for (int k = 0; k < 5; k++) gsq = fe_dbl<M>(gsq);
  if (!fe_eq<M>(got, want) || !fe_eq<M>(gsq, wsq)) atomicAdd(bad, 1);
}
constexpr int MM_ITERS = 512;
template <class M, int VAR> __global__ void __launch_bounds__(256) k_rate(uint32_t* o, const uint32_t* in) {
  Fe<M> a = fe_load<M>(in + 16 * threadIdx.x), b = fe_load<M>(in + 16 * threadIdx.x + 8);
  if (VAR == 0) {
    for (int i = 0; i < MM_ITERS; i++) { a = fe_mul<M>(a, b); b = fe_mul<M>(b, a); }
    uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= a.v[j] ^ b.v[j];
    o[blockIdx.x * blockDim.x + threadIdx.x] = r;
  } else {
    F29 x = unpack29<M>(a), y = unpack29<M>(b);
    for (int i = 0; i < MM_ITERS; i++) {
      if (VAR == 1) { x = mul29<M, false>(x, y); y = mul29<M, false>(y, x); }
      else if (VAR == 2) { x = mul29<M, true>(x, x); y = mul29<M, true>(y, y); }
      else { x = mul29<M, false>(x, y); y = add29<M>(y, x); for (int k = 0; k < 8; k++) { y.v[k + 1] += y.v[k] >> 29; y.v[k] &= MASK29; } x = mul29<M, false>(x, y); }   // mul, lazy add + normalise, mul
    }
    uint32_t r = 0; for (int j = 0; j < 9; j++) r ^= x.v[j] ^ y.v[j];
    o[blockIdx.x * blockDim.x + threadIdx.x] = r;
  }
}

static const uint64_t PQ[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t PR[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static uint64_t rs = 88172645463325252ull;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static bool lt(const uint64_t a[4], const uint64_t p[4]) { for (int i = 3; i >= 0; i--) { if (a[i] < p[i]) return true; if (a[i] > p[i]) return false; } return false; }
static void gen(std::vector<uint32_t>& h, int n, const uint64_t p[4]) {
  h.resize((size_t)n * 16);
  for (int i = 0; i < 2 * n; i++) {
    uint64_t v[4]; int mode = i % 16;
    do {
      for (int j = 0; j < 4; j++) v[j] = rnd();
      v[3] &= 0x3fffffffffffffffull;
      if (mode == 1) { for (int j = 0; j < 4; j++) v[j] = p[j]; v[0] -= 1 + (rnd() % 3); }
      if (mode == 2) { v[0] = rnd() % 4; v[1] = v[2] = v[3] = 0; }
      if (mode == 3) { for (int j = 0; j < 4; j++) v[j] = 0xffffffffffffffffull; v[3] = p[3] - 1; }
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
  printf("%-22s waves/SIMD=%d %.3f ms %.3e ops/s\n", name, blocks / 256, ms, tot / (ms * 1e-3));
}
int main() {
  const int n = 1 << 18;
  uint32_t *d_in, *d_bad, *d_o; CK(hipMalloc(&d_in, (size_t)n * 64)); CK(hipMalloc(&d_bad, 4)); CK(hipMalloc(&d_o, 64 << 20));
  std::vector<uint32_t> h; int fails = 0;
  for (int f = 0; f < 2; f++) {
    gen(h, n, f ? PR : PQ);
    CK(hipMemcpy(d_in, h.data(), (size_t)n * 64, hipMemcpyHostToDevice)); CK(hipMemset(d_bad, 0, 4));
    if (f) hipLaunchKernelGGL(k_check<FrP>, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad, n);
    else hipLaunchKernelGGL(k_check<FqP>, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad, n);
    uint32_t bad; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("%s mul29 / sqr29 vs fe_mul: %u mismatches of %d\n", f ? "Fr" : "Fq", bad, n); fails += bad;
  }
  for (int occ : {1, 2, 3, 4, 8}) {
    rate("fe_mul (8x32, asm)", k_rate<FqP, 0>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("mul29 (9x29, C)", k_rate<FqP, 1>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("sqr29", k_rate<FqP, 2>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("mul29+add+norm+mul29", k_rate<FqP, 3>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
  }
  printf(fails ? "FP29 FAIL\n" : "FP29 OK\n");
  return fails != 0;
}
