// The FP64-FMA field product SURVEY 7 lists as an experiment, priced on MI355X before writing it out.
// Scheme (Emmart / Luitjens / Nath, "Faster modular exponentiation using double precision floating point arithmetic on the GPU"):
// 5 limbs x 52 bits held as doubles; a limb product a_i b_j (104 bits) is split by TWO fused multiply-adds,
//      hi = fma(a_i, b_j, 2^104)            (rounds to a multiple of 2^52: the high half, its integer sits in the mantissa bits)
//      lo = fma(a_i, b_j, C - hi)           (exact: the low half, biased so that the mantissa bits are a non-negative integer)
// and the column sums are INTEGER additions of the two bit patterns (a double cannot hold a sum of five 52-bit terms exactly).
// Per limb product: 2 v_fma_f64 + 1 v_add_f64 (C - hi) + 2 64-bit integer adds; a product has 25 of them, the Montgomery reduction
// (q_i = column * p' mod 2^52, then q_i * p) 5 + 25 more: 55 limb products per field product against the 9 x 29-bit layer's
// 171 v_mad_u64_u32 — each of which multiplies AND accumulates in one 4.2-cycle issue slot.
// This file measures that instruction mix (the dependency structure of one product: columns chained through their accumulators,
// independent across columns) next to fe_mulu on the same chip; it is a PRICE, not a validated product: at 55 x (3 FP64 + 2 u64)
// instructions the FP64 route needs ~275 issue slots of 4.2-4.5 cycles against fe_mulu's ~233, before any limb conversion.
#include "../../spartan-bn254_amd/csrc/fp.cuh"
#include <stdio.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;
constexpr int ITERS = 512;

__device__ __forceinline__ void limb_product(double a, double b, unsigned long long& col_lo, unsigned long long& col_hi) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52 + 0x1p51;
  const double hi = __builtin_fma(a, b, C1);
  const double lo = __builtin_fma(a, b, C2 - hi);
  col_hi += (unsigned long long)__double_as_longlong(hi);
  col_lo += (unsigned long long)__double_as_longlong(lo);
}
// the mix of one Montgomery product: 25 + 5 + 25 limb products over ten column accumulators
__global__ void __launch_bounds__(256) k_fp64_mix(const double* in, unsigned long long* out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double a[5], b[5], p[5];
  for (int i = 0; i < 5; i++) { a[i] = in[(t + i) & 1023]; b[i] = in[(t + 7 * i + 3) & 1023]; p[i] = in[(t + 11 * i + 5) & 1023]; }
  unsigned long long c[11];
  for (int k = 0; k < 11; k++) c[k] = t + k;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 5; j++) limb_product(a[i], b[j], c[i + j], c[i + j + 1]);
#pragma unroll
    for (int i = 0; i < 5; i++) {
      unsigned long long dummy = 0;
      limb_product((double)(c[i] & 0xfffffffffffffull), p[0], c[i], dummy);            // the quotient digit of column i (its low product)
      const double q = __longlong_as_double((long long)((c[i] & 0xfffffffffffffull) | 0x4330000000000000ull)) - 0x1p52;
#pragma unroll
      for (int j = 0; j < 5; j++) limb_product(q, p[j], c[i + j], c[i + j + 1]);
    }
    for (int i = 0; i < 5; i++) { a[i] = __longlong_as_double((long long)((c[5 + i] & 0xfffffffffffffull) | 0x4330000000000000ull)) - 0x1p52; c[i] = c[5 + i] >> 52; }   // next product's operand: data dependence as in a chain of products
  }
  unsigned long long s = 0; for (int k = 0; k < 11; k++) s += c[k];
  out[t] = s;
}
__global__ void __launch_bounds__(256) k_mulu(const uint32_t* in, uint32_t* out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fq a = fe_load<FqP>(in + 8 * (t & 1023)), b = fe_load<FqP>(in + 8 * ((t + 77) & 1023));
  for (int it = 0; it < ITERS; it++) a = fe_mulu(a, b);
  fe_store_packed<FqP>(out + 8 * t, a);
}
int main() {
  double* d_in; unsigned long long* d_out; uint32_t* d_w; uint32_t* d_o2;
  CK(hipMalloc(&d_in, 1024 * 8)); CK(hipMalloc(&d_out, (size_t)2048 * 256 * 8)); CK(hipMalloc(&d_w, 1024 * 32)); CK(hipMalloc(&d_o2, (size_t)2048 * 256 * 32));
  double h[1024]; uint32_t w[8192]; uint64_t s = 88172645463325252ull;
  for (int i = 0; i < 1024; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (double)(s >> 12); }
  for (int i = 0; i < 8192; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w[i] = (uint32_t)(s >> 16); if ((i & 7) == 7) w[i] &= 0x0fffffff; }
  CK(hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, w, sizeof w, hipMemcpyHostToDevice));
  for (int occ : {2, 4, 8}) {
    for (int which = 0; which < 2; which++) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int blocks = 256 * occ;
      auto launch = [&]() { if (which == 0) hipLaunchKernelGGL(k_fp64_mix, dim3(blocks), dim3(256), 0, 0, d_in, d_out); else hipLaunchKernelGGL(k_mulu, dim3(blocks), dim3(256), 0, 0, d_w, d_o2); };
      launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      for (int r = 0; r < 3; r++) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      printf("%-34s blocks/CU=%d: %.3f ms  %.3e products/s\n", which == 0 ? "FP64-FMA instruction mix (5 x 52)" : "fe_mulu (9 x 29-bit, shipped)", occ, ms, (double)blocks * 256 * ITERS / (ms * 1e-3));
    }
  }
  return 0;
}
