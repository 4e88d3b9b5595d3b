// What the non-multiplier VALU instructions of the gather-fed mixed addition cost: the accumulate loop of
// k_acc_first / k_comb_rows (acc += table[random index], next point prefetched) in four variants, same table size and indices:
//   base      xyzz_madd as shipped: 64-byte points (8 x 32-bit words per coordinate) unpacked into 9 x 29-bit limbs per addition,
//             infinity tests on the point and on the accumulator every time
//   prelimb   the table holds the 9 x 29-bit limbs themselves (18 words per point in an 80-byte slot, 16-byte aligned: five
//             dwordx4 loads instead of four): no unpack
//   hoist     64-byte points, but no infinity test on the accumulator (the first addition is peeled off) and a one-limb filter
//             in front of the point's infinity test
//   both      prelimb + hoist
// Points are random field elements (not on the curve): timing only.  Reported: additions per second, whole chip.
#include "../../spartan-bn254_amd/csrc/g1.cuh"
#include <stdio.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
using namespace sbn;

__device__ __forceinline__ uint32_t hash32(uint32_t a, uint32_t b) {
  uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
  x ^= x >> 15; x *= 0xC2B2AE35u; x ^= x >> 13;
  return x;
}
// the madd without the accumulator test and with a cheap filter for an infinite point (canonical coordinates: x = y = 0 has limb 0 of both zero)
__device__ __forceinline__ void xyzz_madd_hoisted(XYZZ& acc, const Affine& q_in, bool neg) {
  if (((q_in.x.v[0] | q_in.y.v[0]) == 0) && aff_is_inf(q_in)) return;
  const Fq qy = neg ? fe_negb<FqP, 2>(q_in.y) : q_in.y;
  const Fq U2 = fe_mulu(q_in.x, acc.ZZ), S2 = fe_mulu(qy, acc.ZZZ);
  const Fq P = fe_normu(fe_subb<FqP, 6, 1>(U2, acc.X));
  const Fq R = fe_normu(fe_subb<FqP, 4, 1>(S2, acc.Y));
  if (fe_maybe_zero(P) && fe_is_zero(P)) {
    if (fe_is_zero(R)) { Affine q; q.x = q_in.x; q.y = fe_normu(qy); acc = xyzz_dbl_affine(q); } else acc = xyzz_inf();
    return;
  }
  const Fq PP = fe_squ(P), PPP = fe_mulu(P, PP), Q = fe_mulu(acc.X, PP);
  const Fq X3 = fe_normu(fe_subb<FqP, 4, 3>(fe_squ(R), fe_add_lazy(fe_add_lazy(PPP, Q), Q)));
  Cols cy; cols_zero(cy);
  cols_mac_lazy<FqP>(cy, R, fe_subb<FqP, 6, 1>(Q, X3)); cols_mac_lazy<FqP>(cy, PPP, fe_negb<FqP, 4>(acc.Y));
  acc.X = X3; acc.Y = cols_reduce<FqP>(cy); acc.ZZ = fe_mulu(acc.ZZ, PP); acc.ZZZ = fe_mulu(acc.ZZZ, PPP);
}
template <bool PRELIMB> __device__ __forceinline__ Affine load_point(const uint32_t* __restrict__ table, uint32_t idx) {
  if (!PRELIMB) return aff_load(table + 16 * (size_t)idx);
  const uint4* q = reinterpret_cast<const uint4*>(table + 20 * (size_t)idx);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
  Affine p;
  p.x.v[0] = a.x; p.x.v[1] = a.y; p.x.v[2] = a.z; p.x.v[3] = a.w; p.x.v[4] = b.x; p.x.v[5] = b.y; p.x.v[6] = b.z; p.x.v[7] = b.w; p.x.v[8] = c.x;
  p.y.v[0] = c.y; p.y.v[1] = c.z; p.y.v[2] = c.w; p.y.v[3] = d.x; p.y.v[4] = d.y; p.y.v[5] = d.z; p.y.v[6] = d.w; p.y.v[7] = e.x; p.y.v[8] = e.y;
  return p;
}
constexpr int MADD_N = 256;
template <bool PRELIMB, bool HOIST> __global__ void __launch_bounds__(256) k_madd(const uint32_t* __restrict__ table, uint32_t mask, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ acc = xyzz_inf();
  Affine nxt = load_point<PRELIMB>(table, hash32(tid, 0) & mask);
  int i0 = 0;
  if (HOIST) {   // the first addition only copies the point
    acc.X = nxt.x; acc.Y = nxt.y; acc.ZZ = fe_one<FqP>(); acc.ZZZ = acc.ZZ;
    nxt = load_point<PRELIMB>(table, hash32(tid, 1) & mask);
    i0 = 1;
  }
  for (int i = i0; i < MADD_N; i++) {
    const Affine p = nxt;
    nxt = load_point<PRELIMB>(table, hash32(tid, i + 1) & mask);
    if (HOIST) xyzz_madd_hoisted(acc, p, (i & 7) == 3); else xyzz_madd(acc, p, (i & 7) == 3);
  }
  xyzz_store(out + 32 * (size_t)tid, acc);
}
// 64-byte slots: 8 x 32-bit words per coordinate, below 2^252; 80-byte slots: 9 limbs below 2^29 (top one below 2^20) per coordinate
__global__ void k_fill(uint32_t* t, size_t npts, int prelimb) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (size_t)gridDim.x * blockDim.x) {
    if (!prelimb) { for (int k = 0; k < 16; k++) { uint32_t v = hash32((uint32_t)i, 17u + k); if ((k & 7) == 7) v &= 0x0fffffffu; t[16 * i + k] = v | (k == 0 ? 1u : 0u); } }
    else { for (int k = 0; k < 20; k++) { uint32_t v = hash32((uint32_t)i, 99u + k) & 0x1fffffffu; if (k == 8 || k == 17) v &= 0xfffffu; if (k >= 18) v = 0; t[20 * i + k] = v | (k == 0 ? 1u : 0u); } }
  }
}
template <class F> float timed(F launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
int main() {
  const int logn = 26; const size_t npts = (size_t)1 << logn; const uint32_t mask = (uint32_t)(npts - 1);
  uint32_t *t64, *t80, *out;
  CK(hipMalloc(&t64, npts * 64)); CK(hipMalloc(&t80, npts * 80)); CK(hipMalloc(&out, (size_t)256 * 8 * 256 * 128));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, t64, npts, 0); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, t80, npts, 1); CK(hipDeviceSynchronize());
  for (int occ : {3, 4}) {
    const int blocks = 256 * occ; const double adds = (double)blocks * 256 * MADD_N;
    float ms;
    ms = timed([&] { hipLaunchKernelGGL((k_madd<false, false>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)t64, mask, out); }, 3);
    printf("base     blocks/CU=%d  %.3f ms  %.3e additions/s\n", occ, ms, adds / (ms * 1e-3));
    ms = timed([&] { hipLaunchKernelGGL((k_madd<true, false>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)t80, mask, out); }, 3);
    printf("prelimb  blocks/CU=%d  %.3f ms  %.3e additions/s\n", occ, ms, adds / (ms * 1e-3));
    ms = timed([&] { hipLaunchKernelGGL((k_madd<false, true>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)t64, mask, out); }, 3);
    printf("hoist    blocks/CU=%d  %.3f ms  %.3e additions/s\n", occ, ms, adds / (ms * 1e-3));
    ms = timed([&] { hipLaunchKernelGGL((k_madd<true, true>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)t80, mask, out); }, 3);
    printf("both     blocks/CU=%d  %.3f ms  %.3e additions/s\n", occ, ms, adds / (ms * 1e-3));
  }
  return 0;
}
