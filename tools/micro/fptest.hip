// On-device check of the shipped field layer (spartan-bn254_amd/csrc/fp.cuh: 9 x 29-bit signed lazy limbs, R = 2^261) against
// the round-1 layer (tools/micro/legacy/fp32.cuh: 8 x 32-bit canonical limbs, R = 2^256) as an independent implementation,
// plus throughput of both.  Every comparison is made on canonical plain integers (what leaves the library).
#include "../../spartan-bn254_amd/csrc/fp.cuh"
#include "legacy/fp32.cuh"
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;

template <class M> struct Leg;
template <> struct Leg<FqP> { typedef sbn32::FqP T; };
template <> struct Leg<FrP> { typedef sbn32::FrP T; };

// plain canonical 8 words -> the two Montgomery domains
template <class M> __device__ Fe<M> to_new(const uint32_t w[8]) { return fe_to_mont<M>(fe_unpack<M>(w)); }
template <class M> __device__ sbn32::Fe<typename Leg<M>::T> to_old(const uint32_t w[8]) {
  sbn32::Fe<typename Leg<M>::T> a; for (int i = 0; i < 8; i++) a.v[i] = w[i]; return sbn32::fe_to_mont(a);
}
template <class M> __device__ bool same(const Fe<M>& x /* lazy, Montgomery */, const sbn32::Fe<typename Leg<M>::T>& y /* Montgomery */) {
  uint32_t w[8]; fe_pack<M>(fe_from_mont<M>(x), w);
  const sbn32::Fe<typename Leg<M>::T> p = sbn32::fe_from_mont(y);
  bool ok = true; for (int i = 0; i < 8; i++) ok = ok && (w[i] == p.v[i]);
  return ok;
}
#define BAD(n) atomicAdd(bad + (n), 1u)
template <class M> __global__ void k_check(const uint32_t* in, uint32_t* bad, int n) {
  typedef typename Leg<M>::T LM;
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  const uint32_t* wa = in + 16 * (size_t)i; const uint32_t* wb = wa + 8;
  uint32_t A[8], B[8]; for (int k = 0; k < 8; k++) { A[k] = wa[k]; B[k] = wb[k]; }
  const Fe<M> a = to_new<M>(A), b = to_new<M>(B);
  const sbn32::Fe<LM> oa = to_old<M>(A), ob = to_old<M>(B);
  if (!same<M>(fe_mul<M>(a, b), sbn32::fe_mul(oa, ob))) BAD(0);
  if (!same<M>(fe_sqr<M>(a), sbn32::fe_mul(oa, oa))) BAD(1);
  if (!same<M>(fe_add<M>(a, b), sbn32::fe_add(oa, ob))) BAD(2);
  if (!same<M>(fe_sub<M>(a, b), sbn32::fe_sub(oa, ob))) BAD(3);
  if (!same<M>(fe_neg<M>(a), sbn32::fe_neg(oa))) BAD(4);
  if (!same<M>(fe_dbl<M>(a), sbn32::fe_dbl(oa))) BAD(5);
  // lazy chains as the hot formulas use them: (a - b)^2, (a - b)(b - a - a), differences of products
  const Fe<M> d = fe_sub_lazy<M>(a, b); const sbn32::Fe<LM> od = sbn32::fe_sub(oa, ob);
  if (!same<M>(fe_sqr<M>(d), sbn32::fe_mul(od, od))) BAD(6);
  const Fe<M> t3 = fe_norm(fe_sub_lazy<M>(fe_sub_lazy<M>(fe_sub_lazy<M>(fe_sqr<M>(d), fe_mul<M>(a, b)), a), a));
  const sbn32::Fe<LM> ot3 = sbn32::fe_sub(sbn32::fe_sub(sbn32::fe_sub(sbn32::fe_mul(od, od), sbn32::fe_mul(oa, ob)), oa), oa);
  if (!same<M>(t3, ot3)) BAD(7);
  if (!same<M>(fe_mul<M>(d, fe_sub_lazy<M>(b, t3)), sbn32::fe_mul(od, sbn32::fe_sub(ob, ot3)))) BAD(8);
  // memory formats: canonical store, table store, unpack
  uint32_t w1[8], w2[8];
  fe_pack<M>(fe_canon<M>(t3), w1);
  { bool c = fe_is_canonical<M>(w1); if (!c) BAD(9); }
  fe_pack<M>(fe_fix_tab<M>(fe_norm(fe_sub_lazy<M>(a, b))), w2);              // (-p, p) -> non-negative representative
  if (!same<M>(fe_unpack<M>(w2), sbn32::fe_sub(oa, ob))) BAD(10);
  if (!same<M>(fe_unpack<M>(w1), ot3)) BAD(11);
  // zero / equality modulo p on non-canonical representatives
  const Fe<M> z1 = fe_sub_lazy<M>(fe_add_lazy<M>(a, b), fe_add_lazy<M>(b, a));
  Fe<M> pm; for (int k = 0; k < NL; k++) pm.v[k] = 3u * p29<M>(k);             // 3p, lazy
  if (!fe_is_zero<M>(z1) || !fe_maybe_zero<M>(z1)) BAD(12);
  if (!fe_is_zero<M>(pm) || !fe_maybe_zero<M>(pm) || !fe_maybe_zero<M>(fe_neg_lazy<M>(pm))) BAD(13);
  if (!fe_eq<M>(fe_add_lazy<M>(a, fe_norm(pm)), a)) BAD(14);
  { bool az = true; for (int k = 0; k < 8; k++) az = az && (A[k] == B[k]); if (fe_eq<M>(a, b) != az) BAD(15); }
  // ark-ff's Montgomery limbs (R = 2^256) are exactly the legacy layer's in-memory form
  uint32_t wark[8]; for (int k = 0; k < 8; k++) wark[k] = oa.v[k];
  if (!same<M>(fe_from_ark_mont<M>(fe_unpack<M>(wark)), oa)) BAD(16);
  { uint32_t wp[8]; fe_pack<M>(fe_ark_mont_to_plain<M>(fe_unpack<M>(wark)), wp); bool ok = true; for (int k = 0; k < 8; k++) ok = ok && wp[k] == A[k]; if (!ok) BAD(17); }
}
template <class M> __global__ void k_inv(const uint32_t* in, uint32_t* bad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  uint32_t A[8]; for (int k = 0; k < 8; k++) A[k] = in[16 * (size_t)i + k];
  const Fe<M> a = to_new<M>(A);
  if (fe_is_zero<M>(a)) return;
  const Fe<M> ai = fe_inv<M>(a), one = fe_mul<M>(a, ai);
  if (!fe_eq<M>(one, fe_one<M>())) atomicAdd(bad, 1u);
}
// Headroom of the 64-bit product columns (fp.cuh: Cols): the documented capacity is 6 products between carry passes + the reduction's own
// 9 x 2^58; the worst case is every operand limb = 2^29 - 1.  6 products, a carry pass, 6 more, ONE reduction must equal twelve single
// products added up (any column overflow would change the value).  The callers' pattern (sumcheck_comb_kernels.cuh: `pend == 6`).
template <class M> __global__ void k_cols_worst(uint32_t* bad) {
  Fe<M> a; for (int k = 0; k < NL; k++) a.v[k] = LMASK;                       // every limb at its maximum ...
  a.v[NL - 1] = 0x007fffffu;                                                  // ... except the top one: a VALUE below 2^255 (table values are below 2.5 r; the sum must fit 261 bits after the reduction)
  Cols s; cols_zero(s);
  for (int i = 0; i < 6; i++) cols_mac<M>(s, a, a);
  cols_carry(s);
  for (int i = 0; i < 6; i++) cols_mac<M>(s, a, a);
  const Fe<M> got = cols_reduce<M>(s);
  const Fe<M> one = fe_mulu<M>(a, a);
  Fe<M> want = fe_zero<M>();
  for (int i = 0; i < 12; i++) want = fe_reduce<M>(fe_add<M>(want, one));
  if (!fe_eq<M>(got, want)) atomicAdd(bad, 1u);
  // and the two-product forms with one lazy operand (limbs below 2^30.6: cols_mac_lazy's documented bound), as xyzz_madd's Y3 uses them
  Fe<M> b; for (int k = 0; k < NL; k++) b.v[k] = 0x5fffffffu;                 // just below 2^30.6
  b.v[NL - 1] = 0x003fffffu;
  Cols t; cols_zero(t);
  cols_mac_lazy<M>(t, a, b); cols_mac_lazy<M>(t, a, b);
  const Fe<M> got2 = cols_reduce<M>(t);
  const Fe<M> bn = fe_normu<M>(b);
  const Fe<M> w1 = fe_mulu<M>(a, bn);
  if (!fe_eq<M>(got2, fe_reduce<M>(fe_add<M>(w1, w1)))) atomicAdd(bad + 1, 1u);
}
constexpr int MM_ITERS = 512;
template <class M, int VAR> __global__ void __launch_bounds__(256) k_rate(uint32_t* o, const uint32_t* in) {
  typedef typename Leg<M>::T LM;
  uint32_t A[8], B[8]; for (int k = 0; k < 8; k++) { A[k] = in[16 * threadIdx.x + k]; B[k] = in[16 * threadIdx.x + 8 + k]; }
  uint32_t r = 0;
  if (VAR == 0) {
    sbn32::Fe<LM> a = to_old<M>(A), b = to_old<M>(B);
    for (int i = 0; i < MM_ITERS; i++) { a = sbn32::fe_mul(a, b); b = sbn32::fe_mul(b, a); }
    for (int j = 0; j < 8; j++) r ^= a.v[j] ^ b.v[j];
  } else if (VAR == 4) {
    sbn32::Fe<LM> a = to_old<M>(A), b = to_old<M>(B);
    for (int i = 0; i < MM_ITERS * 4; i++) { a = sbn32::fe_add(a, b); b = sbn32::fe_sub(b, a); }
    for (int j = 0; j < 8; j++) r ^= a.v[j] ^ b.v[j];
  } else {
    Fe<M> a = to_new<M>(A), b = to_new<M>(B);
    if (VAR == 1) for (int i = 0; i < MM_ITERS; i++) { a = fe_mul<M>(a, b); b = fe_mul<M>(b, a); }
    if (VAR == 2) for (int i = 0; i < MM_ITERS; i++) { a = fe_sqr<M>(a); b = fe_sqr<M>(b); }
    if (VAR == 3) for (int i = 0; i < MM_ITERS * 4; i++) { a = fe_add<M>(a, b); b = fe_sub<M>(b, a); if ((i & 63) == 63) { a = fe_reduce<M>(a); b = fe_reduce<M>(b); } }
    if (VAR == 5) for (int i = 0; i < MM_ITERS; i++) { a = fe_mul<M>(fe_sub_lazy<M>(a, b), b); b = fe_mul<M>(fe_sub_lazy<M>(b, a), a); }      // product of a lazy difference
    for (int j = 0; j < NL; j++) r ^= a.v[j] ^ b.v[j];
  }
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
      if (mode == 5 && (i & 1)) { for (int j = 0; j < 4; j++) v[j] = ((uint64_t)h[(size_t)(i - 1) * 8 + 2 * j + 1] << 32) | h[(size_t)(i - 1) * 8 + 2 * j]; }   // a == b
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
  printf("%-26s waves/SIMD=%d %.3f ms %.3e ops/s\n", name, blocks / 256, ms, tot / (ms * 1e-3));
}
int main() {
  const int n = 1 << 18;
  uint32_t *d_in, *d_bad, *d_o; CK(hipMalloc(&d_in, (size_t)n * 64)); CK(hipMalloc(&d_bad, 128)); CK(hipMalloc(&d_o, 64 << 20));
  std::vector<uint32_t> h; int fails = 0;
  static const char* names[18] = {"mul", "sqr", "add", "sub", "neg", "dbl", "sqr(lazy diff)", "3-term lazy sum", "mul(lazy, lazy diff)", "canonical store", "table store", "unpack",
                                  "is_zero(a+b-b-a)", "is_zero(3p)", "eq(a+3p, a)", "eq(a, b)", "from ark-mont", "ark-mont -> plain"};
  for (int f = 0; f < 2; f++) {
    gen(h, n, f ? PR : PQ);
    CK(hipMemcpy(d_in, h.data(), (size_t)n * 64, hipMemcpyHostToDevice)); CK(hipMemset(d_bad, 0, 128));
    if (f) hipLaunchKernelGGL(k_check<FrP>, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad, n);
    else hipLaunchKernelGGL(k_check<FqP>, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad, n);
    uint32_t bad[32]; CK(hipMemcpy(bad, d_bad, 128, hipMemcpyDeviceToHost));
    for (int k = 0; k < 18; k++) { if (bad[k]) printf("%s %-22s %u mismatches of %d\n", f ? "Fr" : "Fq", names[k], bad[k], n); fails += bad[k]; }
    printf("%s: 18 checks x %d cases vs the legacy 8x32 layer: %s\n", f ? "Fr" : "Fq", n, fails ? "MISMATCH" : "ok");
    CK(hipMemset(d_bad, 0, 128));
    if (f) hipLaunchKernelGGL(k_inv<FrP>, dim3(4096 / 256), dim3(256), 0, 0, d_in, d_bad, 4096);
    else hipLaunchKernelGGL(k_inv<FqP>, dim3(4096 / 256), dim3(256), 0, 0, d_in, d_bad, 4096);
    CK(hipMemcpy(bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("%s inv check: %u mismatches of 4096\n", f ? "Fr" : "Fq", bad[0]); fails += bad[0];
  }
  {
    CK(hipMemset(d_bad, 0, 128));
    hipLaunchKernelGGL(k_cols_worst<FqP>, dim3(1), dim3(1), 0, 0, d_bad); hipLaunchKernelGGL(k_cols_worst<FrP>, dim3(1), dim3(1), 0, 0, d_bad + 2);
    uint32_t bad[4]; CK(hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost));
    printf("Cols headroom, all limbs 2^29 - 1, 6 + 6 products and one reduction: Fq %s, Fr %s; two lazy products (limbs < 2^30.6): Fq %s, Fr %s\n",
           bad[0] ? "OVERFLOW" : "ok", bad[2] ? "OVERFLOW" : "ok", bad[1] ? "OVERFLOW" : "ok", bad[3] ? "OVERFLOW" : "ok");
    fails += bad[0] + bad[1] + bad[2] + bad[3];
  }
  for (int occ : {1, 2, 3, 4, 8}) {      // occ = waves per SIMD (256-thread blocks per CU)
    rate("fe_mul (legacy 8x32)", k_rate<FqP, 0>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("fe_mul (9x29)", k_rate<FqP, 1>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("fe_mul(lazy diff, .)", k_rate<FqP, 5>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("fe_sqr (9x29)", k_rate<FqP, 2>, 256 * occ, 2.0 * MM_ITERS, d_o, d_in);
    rate("fe_add+sub (legacy)", k_rate<FqP, 4>, 256 * occ, 8.0 * MM_ITERS, d_o, d_in);
    rate("fe_add+sub (9x29, safe)", k_rate<FqP, 3>, 256 * occ, 8.0 * MM_ITERS, d_o, d_in);
  }
  printf(fails ? "FPTEST FAIL\n" : "FPTEST OK\n");
  return fails != 0;
}
