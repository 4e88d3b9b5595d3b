// gather.hip — rate of random 64-byte gathers (one affine point) from tables of growing size: does a fixed-base lookup
// table of tens of GB still feed 1.2e10 points/s (the mixed-addition ceiling)?  Each lane issues dependent-free loads with a
// cheap hash as index; 4 loads in flight per lane; result XOR-folded so nothing is optimised away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_fill(uint4* t, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) t[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ t, uint64_t entries, int iters, uint32_t* __restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint64_t h = mix(tid + 1);
  for (int it = 0; it < iters; it += 4) {
    uint64_t i0 = mix(h) % entries, i1 = mix(h + 1) % entries, i2 = mix(h + 2) % entries, i3 = mix(h + 3) % entries; h += 4;
    const uint4* p0 = t + 4 * i0; const uint4* p1 = t + 4 * i1; const uint4* p2 = t + 4 * i2; const uint4* p3 = t + 4 * i3;
    uint4 a0 = p0[0], a1 = p0[1], a2 = p0[2], a3 = p0[3];
    uint4 b0 = p1[0], b1 = p1[1], b2 = p1[2], b3 = p1[3];
    uint4 c0 = p2[0], c1 = p2[1], c2 = p2[2], c3 = p2[3];
    uint4 d0 = p3[0], d1 = p3[1], d2 = p3[2], d3 = p3[3];
    acc.x ^= a0.x ^ a1.x ^ a2.x ^ a3.x ^ b0.x ^ b1.x ^ b2.x ^ b3.x; acc.y ^= c0.x ^ c1.x ^ c2.x ^ c3.x ^ d0.x ^ d1.x ^ d2.x ^ d3.x;
  }
  if ((acc.x ^ acc.y) == 0x12345678u) out[0] = acc.x;
}
int main(int argc, char** argv) {
  uint32_t* out; CK(hipMalloc(&out, 64));
  for (int ai = 1; ai < argc; ai++) {
    const double gb = atof(argv[ai]);
    const uint64_t entries = (uint64_t)(gb * (1ull << 30) / 64);
    uint4* t; if (hipMalloc(&t, entries * 64) != hipSuccess) { printf("%.0f GiB: hipMalloc failed\n", gb); continue; }
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, t, entries * 4); CK(hipDeviceSynchronize());
    const int iters = 256; const unsigned blocks = 256 * 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, t, entries, 16, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, t, entries, iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double g = (double)blocks * 256 * iters;
    printf("table %.0f GiB: %.3e random 64-B gathers/s (%.0f GB/s)\n", gb, g / (ms * 1e-3), g * 64 / (ms * 1e-3) / 1e9);
    CK(hipFree(t));
  }
  return 0;
}
