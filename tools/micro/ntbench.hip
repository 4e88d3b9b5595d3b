// The sumcheck round's memory pattern (per table: 4 element reads at i, i+q, i+2q, i+3q and 2 element writes at i, i+q; 24 tables of 2^21 32-byte
// elements; no arithmetic) with and without the non-temporal hint on the stores / on loads and stores (global_load/store ... nt): does telling the
// memory system that nothing is re-used lift the 4.4 - 4.9 TB/s this pattern reaches?  Cold: a 6 GiB memset between launches.
//   hipcc --offload-arch=gfx950 -O3 -o ntbench tools/micro/ntbench.hip && ./ntbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT_MAX = 24;
struct Tabs { const u32x4* src[NT_MAX]; u32x4* dst[NT_MAX]; int n; };

template <int NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { if (NT >= 2) return __builtin_nontemporal_load(p); return *p; }
template <int NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT >= 1) __builtin_nontemporal_store(v, p); else *p = v; }

template <int NT>
__global__ void __launch_bounds__(256, 2) k_pattern(Tabs t, size_t q, uint32_t* __restrict__ sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll 1
    for (int k = 0; k < t.n; k++) {
      const u32x4* src = t.src[k]; u32x4* dst = t.dst[k];
      u32x4 v[8];
#pragma unroll
      for (int e = 0; e < 4; e++) { v[2 * e] = ld<NT>(src + 2 * (i + e * q)); v[2 * e + 1] = ld<NT>(src + 2 * (i + e * q) + 1); }
      const u32x4 lo0 = v[0] ^ v[4], lo1 = v[1] ^ v[5], hi0 = v[2] ^ v[6], hi1 = v[3] ^ v[7];
      st<NT>(dst + 2 * i, lo0); st<NT>(dst + 2 * i + 1, lo1);
      st<NT>(dst + 2 * (i + q), hi0); st<NT>(dst + 2 * (i + q) + 1, hi1);
      acc ^= lo0 ^ hi1;
    }
  }
  if (acc.x == 0x12345678u && acc.y == 77u) sink[threadIdx.x] = acc.z;
}
// reference points: read everything once / write everything once
template <int NT>
__global__ void __launch_bounds__(256, 2) k_read(Tabs t, size_t q, uint32_t* __restrict__ sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll 1
    for (int k = 0; k < t.n; k++) {
      const u32x4* src = t.src[k];
#pragma unroll
      for (int e = 0; e < 4; e++) acc ^= ld<NT>(src + 2 * (i + e * q)) ^ ld<NT>(src + 2 * (i + e * q) + 1);
    }
  }
  if (acc.x == 0x12345678u && acc.y == 77u) sink[threadIdx.x] = acc.z;
}

template <typename K> static void run(const char* name, K kern, const Tabs& t, size_t q, unsigned gx, uint32_t* sink, double bytes) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  static void* junk = nullptr; if (!junk) CK(hipMalloc(&junk, (size_t)6 << 30));
  float sum = 0;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipMemsetAsync(junk, rep, (size_t)6 << 30, 0));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(256), 0, 0, t, q, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) sum += ms;
  }
  printf("%-34s gx=%4u: COLD %8.1f us  %6.2f TB/s\n", name, gx, sum / 3 * 1e3, bytes / (sum / 3 * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int logn = argc > 1 ? atoi(argv[1]) : 21;
  const size_t n = (size_t)1 << logn, q = n / 4;
  Tabs t; t.n = NT_MAX;
  for (int k = 0; k < NT_MAX; k++) { void* s; void* d; CK(hipMalloc(&s, n * 32)); CK(hipMalloc(&d, n * 16)); CK(hipMemset(s, k + 1, n * 32)); t.src[k] = (const u32x4*)s; t.dst[k] = (u32x4*)d; }
  uint32_t* sink; CK(hipMalloc(&sink, 4096));
  const double rw = (double)NT_MAX * n * 48, ro = (double)NT_MAX * n * 32;
  for (unsigned gx : {512u, 1024u, 2048u}) {
    run("pattern: plain", k_pattern<0>, t, q, gx, sink, rw);
    run("pattern: nt stores", k_pattern<1>, t, q, gx, sink, rw);
    run("pattern: nt loads + stores", k_pattern<2>, t, q, gx, sink, rw);
    run("read only: plain", k_read<0>, t, q, gx, sink, ro);
    run("read only: nt loads", k_read<2>, t, q, gx, sink, ro);
  }
  return 0;
}
