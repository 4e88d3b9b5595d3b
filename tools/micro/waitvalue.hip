// A round trip with the NEXT kernel already queued behind a stream wait (hipStreamWaitValue32, executed by the command processor): the
// host releases it by writing one word instead of paying a launch.  Compared in one run with the plain launch-per-round loop.
//   hipcc --offload-arch=gfx950 -O2 -o waitvalue tools/micro/waitvalue.hip && ./waitvalue [rounds]
// Every host wait is bounded; before the final synchronise the gate word is set past every queued wait, so the stream always drains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using clk = std::chrono::steady_clock;

__global__ void k_answer(volatile uint32_t* mbox, const uint32_t* rslot, uint32_t seq) {
  if (threadIdx.x == 0) {
    const uint32_t r = rslot ? __hip_atomic_load(rslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;     // the "challenge" the host wrote before releasing
    __hip_atomic_store((uint32_t*)mbox + 1, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store((uint32_t*)mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static bool wait_eq(volatile uint32_t* p, uint32_t v) {
  auto t0 = clk::now();
  while (*p != v) if (clk::now() - t0 > std::chrono::milliseconds(300)) return false;
  std::atomic_thread_fence(std::memory_order_acquire);
  return true;
}

int main(int argc, char** argv) {
  const uint32_t rounds = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000;
  uint32_t* hm; CHK(hipHostMalloc((void**)&hm, 4096, hipHostMallocMapped | hipHostMallocCoherent)); memset(hm, 0, 4096);
  volatile uint32_t* mbox = hm; volatile uint32_t* rslot = hm + 64;
  hipStream_t st; CHK(hipStreamCreate(&st));
  // warm-up + plain loop
  for (int pass = 0; pass < 2; pass++) {
    mbox[0] = 0;
    auto t0 = clk::now(); uint32_t ok = 0;
    for (uint32_t i = 1; i <= rounds; i++) {
      *rslot = i * 7u;
      hipLaunchKernelGGL(k_answer, dim3(1), dim3(64), 0, st, mbox, (const uint32_t*)rslot, i);
      if (!wait_eq(mbox, i)) break;
      ok = i;
    }
    CHK(hipStreamSynchronize(st));
    if (pass) printf("launch per round: %u rounds, %.2f us per round\n", ok, std::chrono::duration<double, std::micro>(clk::now() - t0).count() / (ok ? ok : 1));
  }
  // gated loop: the wait + kernel of round i+1 are enqueued before round i is released
  for (int where = 0; where < 2; where++) {
    uint32_t* gate = nullptr;
    if (where == 0) gate = hm + 128;
    else { hipError_t e = hipExtMallocWithFlags((void**)&gate, 8, hipMallocSignalMemory); if (e != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); continue; } }
    *(volatile uint32_t*)gate = 0; mbox[0] = 0;
    auto enqueue = [&](uint32_t i) -> hipError_t {
      hipError_t e = hipStreamWaitValue32(st, gate, i, hipStreamWaitValueGte, 0xffffffffu);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_answer, dim3(1), dim3(64), 0, st, mbox, (const uint32_t*)rslot, i);
      return hipGetLastError();
    };
    hipError_t e = enqueue(1);
    if (e != hipSuccess) { printf("hipStreamWaitValue32 (%s): %s\n", where ? "signal memory" : "pinned host memory", hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    auto t0 = clk::now(); uint32_t ok = 0, bad = 0;
    for (uint32_t i = 1; i <= rounds; i++) {
      *rslot = i * 7u;
      std::atomic_thread_fence(std::memory_order_release);
      *(volatile uint32_t*)gate = i;                         // release round i
      if (i < rounds && enqueue(i + 1) != hipSuccess) break;  // ... and queue round i+1 behind its own wait while round i runs
      if (!wait_eq(mbox, i)) break;
      if (mbox[1] != i * 7u) bad++;
      ok = i;
    }
    const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
    *(volatile uint32_t*)gate = 0xfffffff0u;                 // past every queued wait: the stream drains whatever happened above
    CHK(hipStreamSynchronize(st));
    printf("gated by hipStreamWaitValue32 on %s: %u rounds, %.2f us per round, %u stale challenge reads\n", where ? "signal memory" : "pinned host memory", ok, us / (ok ? ok : 1), bad);
  }
  return 0;
}
