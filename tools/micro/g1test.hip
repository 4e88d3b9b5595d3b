// Consistency of the G1 formulas (g1.cuh) on the device: k*G for k = 1..N by (1) repeated mixed addition, (2) repeated full
// XYZZ addition of a non-trivial representative of G, (3) double-and-add, (4) through the 128-byte memory format each step.
#include "../../spartan-bn254_amd/csrc/g1.cuh"
#include <stdio.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;
constexpr int N = 200;
__device__ bool aff_same(const Affine& a, const Affine& b) { return fe_eq(a.x, b.x) && fe_eq(a.y, b.y); }
__global__ void k_g1(uint32_t* res, uint32_t* scratch) {
  if (threadIdx.x || blockIdx.x) return;
  Affine G; G.x = fe_one<FqP>(); G.y = fe_dbl(G.x);
  // a representative of G with ZZ != 1: (x*4, y*8, 4, 8)  (ZZ = 2^2, ZZZ = 2^3)
  XYZZ Gz; { Fq two = fe_dbl(fe_one<FqP>()), four = fe_dbl(two), eight = fe_dbl(four); Gz.X = fe_mul(G.x, four); Gz.Y = fe_mul(G.y, eight); Gz.ZZ = four; Gz.ZZZ = eight; }
  XYZZ a1 = xyzz_inf(), a2 = xyzz_inf(), a4 = xyzz_inf();
  for (int k = 1; k <= N; k++) {
    xyzz_madd(a1, G, false);
    a2 = xyzz_add_inl(a2, Gz);
    a4 = xyzz_load(scratch); if (k == 1) a4 = xyzz_inf();
    xyzz_madd(a4, G, false); xyzz_store(scratch, a4);
    XYZZ a3 = xyzz_inf();
    for (int b = 31; b >= 0; b--) { a3 = xyzz_dbl(a3); if ((k >> b) & 1) xyzz_madd(a3, G, false); }
    const Affine r1 = xyzz_to_affine(a1), r2 = xyzz_to_affine(a2), r3 = xyzz_to_affine(a3), r4 = xyzz_to_affine(a4);
    res[4 * k + 0] = aff_same(r1, r3); res[4 * k + 1] = aff_same(r2, r3); res[4 * k + 2] = aff_same(r4, r3);
    // subtraction back: (k G) - G == (k-1) G
    XYZZ m = a3; xyzz_madd(m, G, true);
    XYZZ p3 = xyzz_inf(); for (int b = 31; b >= 0; b--) { p3 = xyzz_dbl(p3); if (((k - 1) >> b) & 1) xyzz_madd(p3, G, false); }
    res[4 * k + 3] = (k == 1) ? xyzz_is_inf(m) : aff_same(xyzz_to_affine(m), xyzz_to_affine(p3));
  }
}
// block_sum_quad (quad-cooperative additions) against (sum of the lanes' multipliers) * G.  Lane l holds m_l * G, m_l from `pattern`:
//   0: l + 1 (all distinct)   1: 7 for every lane (every addition is a doubling)   2: +-(l/2 + 1) alternating (every pair cancels)
//   3: non-zero only in lanes = 5 mod 17 (most operands are the identity)   4: l % 4 + 1 (equal points inside and across quads)
__device__ XYZZ small_mul(int m) {            // m * G, m may be negative or zero
  Affine G; G.x = fe_one<FqP>(); G.y = fe_dbl(G.x);
  XYZZ a = xyzz_inf(); const int am = m < 0 ? -m : m;
  for (int b = 30; b >= 0; b--) { a = xyzz_dbl(a); if ((am >> b) & 1) xyzz_madd(a, G, m < 0); }
  return a;
}
__device__ int pattern_m(int pattern, int l) {
  switch (pattern) {
    case 0: return l + 1;
    case 1: return 7;
    case 2: return (l & 1) ? -(l / 2 + 1) : (l / 2 + 1);
    case 3: return (l % 17 == 5) ? 1000 + l : 0;
    default: return l % 4 + 1;
  }
}
__global__ void k_quad(uint32_t* res, int pattern) {
  __shared__ uint32_t sm[4][32];
  const int l = threadIdx.x;
  const XYZZ v = small_mul(pattern_m(pattern, l));
  const XYZZ s = block_sum_quad(v, sm);
  if (l == 0) {
    long long tot = 0; for (int i = 0; i < (int)blockDim.x; i++) tot += pattern_m(pattern, i);
    const XYZZ e = small_mul((int)tot);
    const bool ok = xyzz_is_inf(e) ? xyzz_is_inf(s) : (!xyzz_is_inf(s) && aff_same(xyzz_to_affine(s), xyzz_to_affine(e)));
    res[0] = ok ? 1u : 0u;
  }
}
int main() {
  {
    uint32_t* r; CK(hipMalloc(&r, 4));
    int bad = 0;
    for (int threads : {64, 128, 192, 256}) for (int pat = 0; pat < 5; pat++) {
      CK(hipMemset(r, 0, 4));
      hipLaunchKernelGGL(k_quad, dim3(1), dim3(threads), 0, 0, r, pat); CK(hipDeviceSynchronize());
      uint32_t h = 0; CK(hipMemcpy(&h, r, 4, hipMemcpyDeviceToHost));
      if (!h) { printf("block_sum_quad: %d lanes, pattern %d: MISMATCH\n", threads, pat); bad++; }
    }
    printf("block_sum_quad: %d of 20 cases wrong\n", bad);
    if (bad) { printf("G1TEST FAIL\n"); return 1; }
  }
  uint32_t *d, *s; CK(hipMalloc(&d, 4 * (N + 1) * 4)); CK(hipMalloc(&s, 256)); CK(hipMemset(d, 0, 4 * (N + 1) * 4));
  hipLaunchKernelGGL(k_g1, dim3(1), dim3(64), 0, 0, d, s); CK(hipDeviceSynchronize());
  static uint32_t h[4 * (N + 1)]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
  const char* nm[4] = {"madd chain", "xyzz_add chain", "madd via memory", "k G - G"};
  int fails = 0;
  for (int j = 0; j < 4; j++) { int first = 0, cnt = 0; for (int k = 1; k <= N; k++) if (!h[4 * k + j]) { if (!first) first = k; cnt++; } printf("%-18s %d mismatches of %d (first at k = %d)\n", nm[j], cnt, N, first); fails += cnt; }
  printf(fails ? "G1TEST FAIL\n" : "G1TEST OK\n");
  return fails != 0;
}
