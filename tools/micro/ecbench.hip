// Ceiling of the bucket-accumulation inner loop: xyzz_madd with operands in registers (no memory in the loop).
#include "../../spartan-bn254_amd/csrc/g1.cuh"
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;
constexpr int ITERS = 256;
__global__ void __launch_bounds__(256) k_madd(const uint32_t* pts, uint32_t* out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Affine p = aff_load(pts + 16 * (t & 1023));
  Affine q = aff_load(pts + 16 * ((t + 77) & 1023));
  XYZZ acc = xyzz_from_affine(q);
  for (int i = 0; i < ITERS; i++) { xyzz_madd(acc, p, (i & 1) != 0); p.x = fe_add(p.x, acc.X); }   // keep p changing so nothing hoists
  xyzz_store(out + 32 * t, acc);
}
// round 2's mixed addition (Y3 as two products with a reduction each), kept here for the before / after of round 3's single-reduction form
__device__ __forceinline__ void xyzz_madd_r02(XYZZ& acc, const Affine& q_in, bool neg) {
  if (aff_is_inf(q_in)) return;
  const Fq qy = neg ? fe_negb<FqP, 2>(q_in.y) : q_in.y;
  if (xyzz_is_inf(acc)) { acc.X = q_in.x; acc.Y = fe_normu(qy); acc.ZZ = fe_one<FqP>(); acc.ZZZ = acc.ZZ; return; }
  const Fq U2 = fe_mulu(q_in.x, acc.ZZ), S2 = fe_mulu(qy, acc.ZZZ);
  const Fq P = fe_normu(fe_subb<FqP, 6, 1>(U2, acc.X));
  const Fq R = fe_normu(fe_subb<FqP, 4, 1>(S2, acc.Y));
  if (fe_maybe_zero(P) && fe_is_zero(P)) {
    if (fe_is_zero(R)) { Affine q; q.x = q_in.x; q.y = fe_normu(qy); acc = xyzz_dbl_affine(q); } else acc = xyzz_inf();
    return;
  }
  const Fq PP = fe_squ(P), PPP = fe_mulu(P, PP), Q = fe_mulu(acc.X, PP);
  const Fq X3 = fe_normu(fe_subb<FqP, 4, 3>(fe_squ(R), fe_add_lazy(fe_add_lazy(PPP, Q), Q)));
  const Fq Y3 = fe_normu(fe_subb<FqP, 2, 1>(fe_mulu(R, fe_subb<FqP, 6, 1>(Q, X3)), fe_mulu(acc.Y, PPP)));
  acc.X = X3; acc.Y = Y3; acc.ZZ = fe_mulu(acc.ZZ, PP); acc.ZZZ = fe_mulu(acc.ZZZ, PPP);
}
__global__ void __launch_bounds__(256) k_madd_r02(const uint32_t* pts, uint32_t* out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Affine p = aff_load(pts + 16 * (t & 1023));
  Affine q = aff_load(pts + 16 * ((t + 77) & 1023));
  XYZZ acc = xyzz_from_affine(q);
  for (int i = 0; i < ITERS; i++) { xyzz_madd_r02(acc, p, (i & 1) != 0); p.x = fe_add(p.x, acc.X); }
  xyzz_store(out + 32 * t, acc);
}
__global__ void __launch_bounds__(256) k_add(const uint32_t* pts, uint32_t* out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ a = xyzz_from_affine(aff_load(pts + 16 * (t & 1023)));
  XYZZ b = xyzz_from_affine(aff_load(pts + 16 * ((t + 77) & 1023)));
  for (int i = 0; i < ITERS; i++) { a = xyzz_add(a, b); b.X = fe_add(b.X, a.X); }
  xyzz_store(out + 32 * t, a);
}
int main() {
  uint32_t *d_p, *d_o; CK(hipMalloc(&d_p, 1024 * 64)); CK(hipMalloc(&d_o, (size_t)2048 * 256 * 128));
  std::vector<uint32_t> h(1024 * 16); uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
  for (int i = 0; i < 1024; i++) { h[16 * i + 7] &= 0x0fffffff; h[16 * i + 15] &= 0x0fffffff; }   // arbitrary field elements (not on the curve: timing only)
  CK(hipMemcpy(d_p, h.data(), 1024 * 64, hipMemcpyHostToDevice));
  for (int occ : {1, 2, 4, 8}) {
    for (int which = 0; which < 3; which++) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      int blocks = 256 * occ;
      auto launch = [&]() { if (which == 0) hipLaunchKernelGGL(k_madd, dim3(blocks), dim3(256), 0, 0, d_p, d_o); else if (which == 1) hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, d_p, d_o); else hipLaunchKernelGGL(k_madd_r02, dim3(blocks), dim3(256), 0, 0, d_p, d_o); };
      launch();
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      for (int r = 0; r < 3; r++) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      double ops = (double)blocks * 256 * ITERS;
      printf("%s blocks/CU=%d: %.3f ms  %.3e ops/s\n", which == 0 ? "xyzz_madd          " : which == 1 ? "xyzz_add           " : "xyzz_madd (round 2)", occ, ms, ops / (ms * 1e-3));
    }
  }
  return 0;
}
