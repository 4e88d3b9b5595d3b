// What a sumcheck round costs when the kernel STAYS on the device and trades (results, challenge) with the host through pinned host memory —
// the "second attempt" protocol of DESIGN.md §4.4 (round 3's resident tail kernel lost to launch-per-round: 22.0 against 17.5 us):
//   * device -> host: every 16-byte piece of a block's result carries the round's sequence number in its last word (12 B data + 4 B seq),
//     so the stores need no drain and no separate flag: the host has the result when all pieces show the sequence number;
//   * host -> device: the challenge travels the same way in ONE 64-byte line (3 pieces), polled by three lanes with one 16-byte load each;
//   * every device wait is bounded by the 100 MHz real-time counter (the kernel gives up after 20 ms without an answer).
// B blocks (one per sumcheck instance), `work` dependent 64-bit multiply-adds per thread between challenge and result (stands for bind + evaluate).
//   hipcc --offload-arch=gfx950 -O2 -o resident tools/micro/resident.hip && ./resident
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <atomic>
#include <emmintrin.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load_sys16(const uint32_t* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_sys16(uint32_t* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// cmd: 3 pieces of 16 B in one line; res: per block 8 pieces of 16 B
__global__ void __launch_bounds__(256) k_resident(const uint32_t* cmd, uint32_t* res, uint32_t rounds, uint32_t work, uint32_t* status) {
  __shared__ uint32_t ch[12];
  __shared__ uint32_t okflag;
  const uint32_t t = threadIdx.x;
  uint32_t done = 0;
  unsigned long long acc = t + 1;
  for (uint32_t s = 1; s <= rounds; s++) {
    if (t < 64) {                       // wave 0 polls: lanes 0..2 one piece each
      const unsigned long long t0 = wall_clock64();
      bool ok = false;
      while (true) {
        u32x4 v = {0, 0, 0, 0};
        if (t < 3) v = load_sys16(cmd + 4 * t);
        const bool mine = t >= 3 || v.w == s;
        if (__all(mine)) { if (t < 3) { ch[4 * t] = v.x; ch[4 * t + 1] = v.y; ch[4 * t + 2] = v.z; } ok = true; break; }
        if (wall_clock64() - t0 > 2000000ull) break;           // 20 ms
      }
      if (t == 0) okflag = ok ? 1u : 0u;
    }
    __syncthreads();
    if (!okflag) break;
    // the round's arithmetic: a dependent chain seeded by the challenge
    unsigned long long x = acc ^ ch[t % 9];
    for (uint32_t i = 0; i < work; i++) x = x * 0x9E3779B97F4A7C15ull + (x >> 29);
    acc = x;
    // wave butterfly + block fold, as the real kernel's sums
    for (int d = 1; d < 64; d <<= 1) x += __shfl_xor(x, d);
    __shared__ unsigned long long ws[4];
    if ((t & 63) == 0) ws[t >> 6] = x;
    __syncthreads();
    if (t < 8) {
      const unsigned long long tot = ws[0] + ws[1] + ws[2] + ws[3];
      store_sys16(res + 32 * blockIdx.x + 4 * t, u32x4{(uint32_t)tot, (uint32_t)(tot >> 32), t, s});
    }
    done = s;
    __syncthreads();
  }
  if (t == 0) status[blockIdx.x] = done;
}

static inline void host_cmd(volatile uint32_t* cmd, uint32_t s, uint32_t seedv) {
  for (int p = 0; p < 3; p++) {
    __m128i v = _mm_set_epi32((int)s, (int)(seedv + 2), (int)(seedv + 1), (int)(seedv + p));
    _mm_store_si128((__m128i*)(cmd + 4 * p), v);
  }
  std::atomic_thread_fence(std::memory_order_release);
}

static int run(uint32_t* hm, uint32_t B, uint32_t rounds, uint32_t work) {
  uint32_t* d_status; CHK(hipMalloc(&d_status, 4 * B)); CHK(hipMemset(d_status, 0, 4 * B));
  memset(hm, 0, 65536);
  volatile uint32_t* cmd = hm; volatile uint32_t* res = hm + 1024;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipStream_t st; CHK(hipStreamCreate(&st));
  CHK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_resident, dim3(B), dim3(256), 0, st, (const uint32_t*)hm, hm + 1024, rounds, work, d_status);
  uint32_t got = 0; uint32_t h = 1;
  host_cmd(cmd, 1, h);
  for (uint32_t s = 1; s <= rounds; s++) {
    auto w0 = std::chrono::steady_clock::now();
    while (true) {
      bool all = true;
      for (uint32_t b = 0; b < B && all; b++) for (int p = 0; p < 8; p++) if (res[32 * b + 4 * p + 3] != s) { all = false; break; }
      if (all) break;
      if (std::chrono::steady_clock::now() - w0 > std::chrono::milliseconds(100)) goto out;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (uint32_t b = 0; b < B; b++) h = h * 31 + res[32 * b];       // stands for combine + transcript
    got = s;
    if (s < rounds) host_cmd(cmd, s + 1, h);
  }
out:
  CHK(hipStreamSynchronize(st));
  auto t1 = std::chrono::steady_clock::now();
  uint32_t status[64]; CHK(hipMemcpy(status, d_status, 4 * B, hipMemcpyDeviceToHost));
  uint32_t mn = status[0]; for (uint32_t b = 1; b < B; b++) if (status[b] < mn) mn = status[b];
  const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
  printf("blocks %2u  work %5u: host saw %u, slowest block completed %u of %u rounds, %.2f us per round (launch included)\n", B, work, got, mn, rounds, us / (mn ? mn : 1));
  hipFree(d_status); hipStreamDestroy(st);
  return 0;
}

// launch-per-round reference with the same arithmetic: one launch + a flag in pinned memory per round
__global__ void __launch_bounds__(256) k_oneshot(uint32_t seedv, uint32_t* res, uint32_t s, uint32_t work) {
  const uint32_t t = threadIdx.x;
  unsigned long long x = (t + 1) ^ seedv;
  for (uint32_t i = 0; i < work; i++) x = x * 0x9E3779B97F4A7C15ull + (x >> 29);
  for (int d = 1; d < 64; d <<= 1) x += __shfl_xor(x, d);
  __shared__ unsigned long long ws[4];
  if ((t & 63) == 0) ws[t >> 6] = x;
  __syncthreads();
  if (t < 8) { const unsigned long long tot = ws[0] + ws[1] + ws[2] + ws[3]; store_sys16(res + 32 * blockIdx.x + 4 * t, u32x4{(uint32_t)tot, (uint32_t)(tot >> 32), t, s}); }
}
static int run_launch(uint32_t* hm, uint32_t B, uint32_t rounds, uint32_t work) {
  memset(hm, 0, 65536);
  volatile uint32_t* res = hm + 1024;
  hipStream_t st; CHK(hipStreamCreate(&st)); CHK(hipDeviceSynchronize());
  uint32_t h = 1;
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t s = 1; s <= rounds; s++) {
    hipLaunchKernelGGL(k_oneshot, dim3(B), dim3(256), 0, st, h, hm + 1024, s, work);
    auto w0 = std::chrono::steady_clock::now();
    while (true) {
      bool all = true;
      for (uint32_t b = 0; b < B && all; b++) for (int p = 0; p < 8; p++) if (res[32 * b + 4 * p + 3] != s) { all = false; break; }
      if (all) break;
      if (std::chrono::steady_clock::now() - w0 > std::chrono::milliseconds(100)) { printf("launch form timed out\n"); return 1; }
    }
    for (uint32_t b = 0; b < B; b++) h = h * 31 + res[32 * b];
  }
  CHK(hipStreamSynchronize(st));
  auto t1 = std::chrono::steady_clock::now();
  printf("blocks %2u  work %5u: one launch per round: %.2f us per round\n", B, work, std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds);
  hipStreamDestroy(st);
  return 0;
}

int main(int argc, char** argv) {
  const uint32_t rounds = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
  uint32_t* hm; CHK(hipHostMalloc((void**)&hm, 65536, hipHostMallocMapped | hipHostMallocCoherent));
  const uint32_t Bs[] = {1, 18, 24}; const uint32_t works[] = {0, 200, 600};
  for (uint32_t B : Bs) for (uint32_t w : works) { if (run(hm, B, rounds, w)) return 1; if (run_launch(hm, B, rounds, w)) return 1; }
  return 0;
}
