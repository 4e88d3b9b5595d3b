// Micro-benchmarks for the integer paths a BN254 modmul can ride on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

constexpr int ITERS = 4096;

__global__ void k_mad64(uint32_t* o, const uint32_t* in) {
  uint32_t x = in[threadIdx.x], y = in[threadIdx.x + 64];
  uint64_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 3, a4 = x+5, a5=y+7, a6=x+9, a7=y+11;
  for (int i = 0; i < ITERS; i++) {
    a0 = (uint64_t)(uint32_t)a0 * y + a0; a1 = (uint64_t)(uint32_t)a1 * x + a1;
    a2 = (uint64_t)(uint32_t)a2 * y + a2; a3 = (uint64_t)(uint32_t)a3 * x + a3;
    a4 = (uint64_t)(uint32_t)a4 * y + a4; a5 = (uint64_t)(uint32_t)a5 * x + a5;
    a6 = (uint64_t)(uint32_t)a6 * y + a6; a7 = (uint64_t)(uint32_t)a7 * x + a7;
  }
  uint64_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  o[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
__global__ void k_mullo(uint32_t* o, const uint32_t* in) {
  uint32_t x = in[threadIdx.x] | 1, y = in[threadIdx.x + 64] | 1;
  uint32_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 3, a4 = x+5, a5=y+7, a6=x+9, a7=y+11;
  for (int i = 0; i < ITERS; i++) {
    a0 *= y; a1 *= x; a2 *= y; a3 *= x; a4 *= y; a5 *= x; a6 *= y; a7 *= x;
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_mulhi(uint32_t* o, const uint32_t* in) {
  uint32_t x = in[threadIdx.x] | 0x80000001u, y = in[threadIdx.x + 64] | 0x80000001u;
  uint32_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 3, a4 = x+5, a5=y+7, a6=x+9, a7=y+11;
  for (int i = 0; i < ITERS; i++) {
    a0 = __umulhi(a0, y) | 0x80000000u; a1 = __umulhi(a1, x)| 0x80000000u; a2 = __umulhi(a2, y)| 0x80000000u; a3 = __umulhi(a3, x)| 0x80000000u;
    a4 = __umulhi(a4, y)| 0x80000000u; a5 = __umulhi(a5, x)| 0x80000000u; a6 = __umulhi(a6, y)| 0x80000000u; a7 = __umulhi(a7, x)| 0x80000000u;
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_mad24(uint32_t* o, const uint32_t* in) {
  uint32_t x = in[threadIdx.x] | 1, y = in[threadIdx.x + 64] | 1;
  uint32_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 3, a4 = x+5, a5=y+7, a6=x+9, a7=y+11;
  for (int i = 0; i < ITERS; i++) {
    a0 = __umul24(a0, y) + a1; a1 = __umul24(a1, x) + a2;
    a2 = __umul24(a2, y) + a3; a3 = __umul24(a3, x) + a4;
    a4 = __umul24(a4, y) + a5; a5 = __umul24(a5, x) + a6;
    a6 = __umul24(a6, y) + a7; a7 = __umul24(a7, x) + a0;
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_fma64(uint32_t* o, const uint32_t* in) {
  double x = 1.0 + 1e-9 * in[threadIdx.x], y = 1e-9 * in[threadIdx.x + 64];
  double a0 = x, a1 = y, a2 = x + 1, a3 = y + 3, a4 = x+5, a5=y+7, a6=x+9, a7=y+11;
  for (int i = 0; i < ITERS; i++) {
    a0 = __builtin_fma(a0, x, y); a1 = __builtin_fma(a1, x, y); a2 = __builtin_fma(a2, x, y); a3 = __builtin_fma(a3, x, y);
    a4 = __builtin_fma(a4, x, y); a5 = __builtin_fma(a5, x, y); a6 = __builtin_fma(a6, x, y); a7 = __builtin_fma(a7, x, y);
  }
  double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  o[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(r);
}
__global__ void k_add(uint32_t* o, const uint32_t* in) {
  uint32_t x = in[threadIdx.x] | 1, y = in[threadIdx.x + 64] | 1;
  uint32_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 3, a4 = x+5, a5=y+7, a6=x+9, a7=y+11;
  for (int i = 0; i < ITERS; i++) {
    a0 = (a0 + y) ^ x; a1 = (a1 + x) ^ y; a2 = (a2 + y)^x; a3 = (a3 + x)^y; a4 = (a4 + y)^x; a5 = (a5 + x)^y; a6 = (a6 + y)^x; a7 = (a7 + x)^y;
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// ---- Montgomery multiplication, 8x32 limbs, no-carry CIOS (p top limb < 2^31) ----
__constant__ uint32_t P_[8] = {0xd87cfd47u,0x3c208c16u,0x6871ca8du,0x97816a91u,0x8181585du,0xb85045b6u,0xe131a029u,0x30644e72u};
#define NINV 0xe4866389u
struct fe { uint32_t v[8]; };
__device__ __forceinline__ fe mont_mul_cios(const fe& a, const fe& b) {
  const uint32_t p[8] = {0xd87cfd47u,0x3c208c16u,0x6871ca8du,0x97816a91u,0x8181585du,0xb85045b6u,0xe131a029u,0x30644e72u};
  uint32_t t[8] = {0,0,0,0,0,0,0,0};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t A = (uint64_t)a.v[0] * b.v[i] + t[0];
    uint32_t m = (uint32_t)A * NINV;
    uint64_t C = (uint64_t)m * p[0] + (uint32_t)A;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      A = (uint64_t)a.v[j] * b.v[i] + t[j] + (A >> 32);
      C = (uint64_t)m * p[j] + (uint32_t)A + (C >> 32);
      t[j - 1] = (uint32_t)C;
    }
    t[7] = (uint32_t)(C >> 32) + (uint32_t)(A >> 32);
  }
  // conditional subtract
  uint32_t r[8]; uint64_t br = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { uint64_t d = (uint64_t)t[j] - p[j] - br; r[j] = (uint32_t)d; br = (d >> 63); }
  fe o;
#pragma unroll
  for (int j = 0; j < 8; j++) o.v[j] = br ? t[j] : r[j];
  return o;
}
constexpr int MM_ITERS = 512;
__global__ void __launch_bounds__(256) k_montmul(uint32_t* o, const uint32_t* in) {
  fe a, b;
  for (int j = 0; j < 8; j++) { a.v[j] = in[(threadIdx.x * 8 + j) & 1023]; b.v[j] = in[(threadIdx.x * 8 + j + 512) & 1023]; }
  a.v[7] &= 0x0fffffffu; b.v[7] &= 0x0fffffffu;
  for (int i = 0; i < MM_ITERS; i++) { a = mont_mul_cios(a, b); b = mont_mul_cios(b, a); }
  uint32_t r = 0; for (int j = 0; j < 8; j++) r ^= a.v[j] ^ b.v[j];
  o[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K>
double run(const char* name, K kern, int blocks, int threads, double ops_per_thread, uint32_t* d_o, uint32_t* d_in) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_o, d_in); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_o, d_in);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  double ops = ops_per_thread * blocks * (double)threads;
  double rate = ops / (ms * 1e-3);
  // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
  double waves_instr = ops / 64.0; double cyc = (ms * 1e-3) * 2.4e9 * 1024.0 / waves_instr;
  printf("%-10s blocks=%5d thr=%d  %.3f ms  %.3e ops/s  ~%.2f cyc/wave-instr/SIMD@2.4GHz\n", name, blocks, threads, ms, rate, cyc);
  return rate;
}

int main() {
  uint32_t *d_o, *d_in; CK(hipMalloc(&d_o, 64 << 20)); CK(hipMalloc(&d_in, 4096 * 4));
  std::vector<uint32_t> h(4096); uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
  CK(hipMemcpy(d_in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s CUs=%d clock=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  for (int occ : {1, 2, 4, 8}) {
    int blocks = 256 * occ; // 256-thread blocks => occ blocks/CU => occ waves/SIMD
    printf("--- %d waves/SIMD ---\n", occ);
    run("add", k_add, blocks, 256, 16.0 * ITERS, d_o, d_in);
    run("mad_u64_u32", k_mad64, blocks, 256, 8.0 * ITERS, d_o, d_in);
    run("mul_lo_u32", k_mullo, blocks, 256, 8.0 * ITERS, d_o, d_in);
    run("mul_hi_u32", k_mulhi, blocks, 256, 16.0 * ITERS, d_o, d_in);
    run("mul_u24+add", k_mad24, blocks, 256, 8.0 * ITERS, d_o, d_in);
    run("fma_f64", k_fma64, blocks, 256, 8.0 * ITERS, d_o, d_in);
    run("montmul", k_montmul, blocks, 256, 2.0 * MM_ITERS, d_o, d_in);
  }
  return 0;
}
