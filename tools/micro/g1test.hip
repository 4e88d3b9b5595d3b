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
int main() {
  uint32_t *d, *s; CK(hipMalloc(&d, 4 * (N + 1) * 4)); CK(hipMalloc(&s, 256)); CK(hipMemset(d, 0, 4 * (N + 1) * 4));
  hipLaunchKernelGGL(k_g1, dim3(1), dim3(64), 0, 0, d, s); CK(hipDeviceSynchronize());
  static uint32_t h[4 * (N + 1)]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
  const char* nm[4] = {"madd chain", "xyzz_add chain", "madd via memory", "k G - G"};
  int fails = 0;
  for (int j = 0; j < 4; j++) { int first = 0, cnt = 0; for (int k = 1; k <= N; k++) if (!h[4 * k + j]) { if (!first) first = k; cnt++; } printf("%-18s %d mismatches of %d (first at k = %d)\n", nm[j], cnt, N, first); fails += cnt; }
  printf(fails ? "G1TEST FAIL\n" : "G1TEST OK\n");
  return fails != 0;
}
