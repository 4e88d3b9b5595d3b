// Host <-> resident kernel ping-pong: what a round trip costs when the kernel publishes into coherent pinned HOST memory and the host
// answers by writing (a) into the same pinned host memory (the kernel polls over PCIe) or (b) straight into fine-grained DEVICE memory
// through the BAR (the kernel polls its own HBM).  One wave; every wait is bounded (the kernel leaves after 50 ms without an answer).
//   hipcc --offload-arch=gfx950 -O2 -o pingpong tools/micro/pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <atomic>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_pingpong(volatile uint32_t* to_host, const uint32_t* from_host, uint32_t rounds, uint32_t* status) {
  if (threadIdx.x != 0) return;
  uint32_t done = 0;
  for (uint32_t s = 1; s <= rounds; s++) {
    __hip_atomic_store((uint32_t*)to_host, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    bool ok = false;
    while (wall_clock64() - t0 < 5000000ull) {                 // 50 ms at 100 MHz
      if (__hip_atomic_load(from_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == s) { ok = true; break; }
    }
    if (!ok) break;
    done = s;
  }
  *status = done;
}

static int run(const char* label, volatile uint32_t* to_host_h, uint32_t* to_host_d, volatile uint32_t* from_host_h, uint32_t* from_host_d, uint32_t rounds) {
  uint32_t* d_status; CHK(hipMalloc(&d_status, 4)); CHK(hipMemset(d_status, 0, 4));
  *to_host_h = 0; *from_host_h = 0;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipStream_t st; CHK(hipStreamCreate(&st));
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, st, to_host_h ? (volatile uint32_t*)to_host_d : nullptr, (const uint32_t*)from_host_d, rounds, d_status);
  uint32_t got = 0;
  for (uint32_t s = 1; s <= rounds; s++) {
    auto w0 = std::chrono::steady_clock::now();
    while (*to_host_h != s) { if (std::chrono::steady_clock::now() - w0 > std::chrono::milliseconds(200)) goto out; }
    std::atomic_thread_fence(std::memory_order_acquire);
    *from_host_h = s;
    std::atomic_thread_fence(std::memory_order_release);
    got = s;
  }
out:
  CHK(hipStreamSynchronize(st));
  auto t1 = std::chrono::steady_clock::now();
  uint32_t status = 0; CHK(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
  const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
  printf("%s: host saw %u, kernel completed %u of %u rounds, %.2f us per round trip\n", label, got, status, rounds, us / (status ? status : 1));
  hipFree(d_status); hipStreamDestroy(st);
  return 0;
}

int main(int argc, char** argv) {
  const uint32_t rounds = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
  uint32_t* hm; CHK(hipHostMalloc((void**)&hm, 4096, hipHostMallocMapped | hipHostMallocCoherent));
  memset(hm, 0, 4096);
  // (a) both directions through pinned host memory
  if (run("answer in pinned host memory (kernel polls over PCIe)", hm, hm, hm + 64, hm + 64, rounds)) return 1;
  // (b) the answer is written by the host into fine-grained device memory
  uint32_t* dm = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&dm, 4096, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e)); return 0; }
  CHK(hipMemset(dm, 0, 4096)); CHK(hipDeviceSynchronize());
  hipPointerAttribute_t at; memset(&at, 0, sizeof at);
  if (hipPointerGetAttributes(&at, dm) == hipSuccess) printf("fine-grained device memory: type %d, hostPointer %p, devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
  printf("writing to it from the host ...\n"); fflush(stdout);
  ((volatile uint32_t*)dm)[128] = 7u;                       // a fault here ends this process only
  printf("host write done, read back %u\n", ((volatile uint32_t*)dm)[128]); fflush(stdout);
  if (run("answer in fine-grained DEVICE memory (kernel polls HBM)", hm, hm, (volatile uint32_t*)dm, dm, rounds)) return 1;
  return 0;
}
