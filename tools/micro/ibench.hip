// Issue cost (cycles per wave-instruction per SIMD) of the VALU instructions the field arithmetic is made of, on gfx950.
// Every test is one inline-asm block of 16 instances on independent registers inside a counted loop; cost = kernel time x
// in-kernel clock x SIMDs / wave-instructions.  The in-kernel clock is measured (s_memtime / s_memrealtime), the chip lowers it
// under load (MI355X_MICROARCH.md, DVFS give-back), so nominal-clock numbers mislead.
// Build: make -C tools/micro ibench ; run on the GPU box: tools/micro/ibench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int ITERS = 2048;

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
// 16 x 32-bit registers r[0..15], 8 x 64-bit accumulators q[0..7], two multiplier operands x, y
#define OPS32 "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
#define OPS64 "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])

template <int T> __global__ void __launch_bounds__(256) k_inst(uint32_t* o, const uint32_t* in, unsigned long long* clk) {
  uint32_t r[16]; uint64_t q[8];
  for (int i = 0; i < 16; i++) r[i] = in[(threadIdx.x + 7 * i) & 1023] | 1u;
  for (int i = 0; i < 8; i++) q[i] = ((uint64_t)r[i] << 32) | r[i + 8];
  uint32_t x = in[threadIdx.x & 1023] | 3u, y = in[(threadIdx.x + 99) & 1023] | 5u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; it++) {
    if (T == 0) asm volatile(
#define X(i) "v_add_u32 %" #i ", %" #i ", %16\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x));
    if (T == 1) asm volatile(
#define X(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %16\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x) : "vcc");
    if (T == 2) asm volatile(     // one 16-long carry chain, as fe_add's
#define X(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %16, vcc\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x) : "vcc");
    if (T == 3) asm volatile(     // the capture form: counter += carry
#define X(i) "v_addc_co_u32 %" #i ", vcc, 0, %" #i ", vcc\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x) : "vcc");
    if (T == 4) asm volatile(
#define X(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x) : "vcc");
    if (T == 5) asm volatile(     // accumulate into 8 independent 64-bit accumulators, twice
#define X(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "v"(y) : "vcc");
    if (T == 6) asm volatile(     // fresh products (addend 0)
#define X(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, 0\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "v"(y) : "vcc");
    if (T == 7) asm volatile(
#define X(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x));
    if (T == 8) asm volatile(     // the pair of fe_mul: mad then capture (8 pairs = 16 instructions)
#define X(i) "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i "\n\tv_addc_co_u32 %8, vcc, 0, %8, vcc\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64, "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "v"(y) : "vcc");
    if (T == 9) asm volatile(     // 64-bit add in one instruction
#define X(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %8\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64 : "v"(q[7]));
    if (T == 10) asm volatile(
#define X(i) "v_lshrrev_b64 %" #i ", 29, %" #i "\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64);
    if (T == 11) asm volatile(
#define X(i) "v_and_b32 %" #i ", %16, %" #i "\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x));
    if (T == 12) asm volatile(
#define X(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 29\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x));
    if (T == 13) asm volatile(
#define X(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x), "v"(y));
    if (T == 14) asm volatile(    // carry-out into a non-VCC SGPR pair
#define X(i) "v_mad_u64_u32 %" #i ", %10, %8, %9, %" #i "\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "v"(y), "s"((uint64_t)0) : "vcc");
    if (T == 15) asm volatile(
#define X(i) "v_mul_hi_u32 %" #i ", %" #i ", %16\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x));
    if (T == 16) asm volatile(
#define X(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x), "v"(y));
    if (T == 17) asm volatile(    // subtract-with-borrow chain (fe_sub / cond_sub)
#define X(i) "v_subb_co_u32 %" #i ", vcc, %" #i ", %16, vcc\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x) : "vcc");
    if (T == 18) asm volatile(    // fresh product with an SGPR multiplier (m * p[k] form)
#define X(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "s"(0x3c208c16u) : "vcc");
    if (T == 19) asm volatile(    // 24-bit multiply high/low pair
#define X(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS32 : "v"(x));
    // dependent chains: the same accumulator every D-th instruction (D = 1, 2, 4) — what a column sum of products looks like
    if (T == 20) asm volatile(
#define X(i) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      R16(X)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "v"(y) : "vcc");
    if (T == 21) asm volatile(
#define X(i) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "v"(y) : "vcc");
    if (T == 22) asm volatile(
#define X(i) "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
      X(0) X(1) X(2) X(3)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "v"(y) : "vcc");
    if (T == 23) asm volatile(    // the carry step between two columns: shift, then add (dependent pair, 8 times)
#define X(i) "v_lshrrev_b64 %0, 29, %0\n\tv_lshl_add_u64 %0, %1, 0, %0\n\t"
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
      "s_nop 0" : OPS64 : "v"(x), "v"(y) : "vcc");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
  for (int i = 0; i < 16; i++) acc ^= r[i];
  for (int i = 0; i < 8; i++) acc ^= (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
  o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int T> void run(const char* name, uint32_t* d_o, uint32_t* d_in, unsigned long long* d_clk) {
  for (int occ : {1, 2, 4, 8}) {
    const int blocks = 256 * occ;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_inst<T>, dim3(blocks), dim3(256), 0, 0, d_o, d_in, d_clk); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_inst<T>, dim3(blocks), dim3(256), 0, 0, d_o, d_in, d_clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz; std::vector<double> cyc;
    for (int b = 0; b < blocks; b++) if (h[2 * b + 1]) { ghz.push_back((double)h[2 * b] / ((double)h[2 * b + 1] * 10.0)); cyc.push_back((double)h[2 * b]); }   // memrealtime ticks at 100 MHz
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double clock = ghz.empty() ? 0 : ghz[ghz.size() / 2];
    // in-kernel: one wave's loop took cyc cycles for ITERS*16 instructions while `occ` waves shared its SIMD
    const double per_inst_simd = cyc.empty() ? 0 : cyc[cyc.size() / 2] / (double)(ITERS * 16) / occ;
    printf("%-34s waves/SIMD=%d  %.3f ms  clock %.2f GHz  %.2f cycles per wave-instruction per SIMD (wall-based %.2f)\n", name, occ, ms, clock, per_inst_simd,
           ms * 1e-3 * clock * 1e9 * 1024.0 / ((double)blocks * 4 * ITERS * 16));
  }
}
int main() {
  uint32_t *d_in, *d_o; unsigned long long* d_clk;
  CK(hipMalloc(&d_in, 4096)); CK(hipMalloc(&d_o, (size_t)2048 * 256 * 4)); CK(hipMalloc(&d_clk, 2048 * 16));
  std::vector<uint32_t> h(1024); uint64_t s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
  CK(hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice));
  run<0>("v_add_u32", d_o, d_in, d_clk);
  run<1>("v_add_co_u32 (carry out)", d_o, d_in, d_clk);
  run<2>("v_addc_co_u32 chain", d_o, d_in, d_clk);
  run<3>("v_addc_co_u32 0,C,vcc (capture)", d_o, d_in, d_clk);
  run<17>("v_subb_co_u32 chain", d_o, d_in, d_clk);
  run<4>("v_cndmask_b32 vcc", d_o, d_in, d_clk);
  run<5>("v_mad_u64_u32 accumulate", d_o, d_in, d_clk);
  run<6>("v_mad_u64_u32 fresh", d_o, d_in, d_clk);
  run<18>("v_mad_u64_u32 sgpr operand", d_o, d_in, d_clk);
  run<14>("v_mad_u64_u32 carry->sgpr pair", d_o, d_in, d_clk);
  run<8>("mad + capture pair (per instr)", d_o, d_in, d_clk);
  run<7>("v_mul_lo_u32", d_o, d_in, d_clk);
  run<15>("v_mul_hi_u32", d_o, d_in, d_clk);
  run<16>("v_mad_u32_u24", d_o, d_in, d_clk);
  run<19>("v_mul_u32_u24", d_o, d_in, d_clk);
  run<20>("v_mad_u64_u32 chain distance 1", d_o, d_in, d_clk);
  run<21>("v_mad_u64_u32 chain distance 2", d_o, d_in, d_clk);
  run<22>("v_mad_u64_u32 chain distance 4", d_o, d_in, d_clk);
  run<23>("shift+add dependent pair chain", d_o, d_in, d_clk);
  run<9>("v_lshl_add_u64 (64-bit add)", d_o, d_in, d_clk);
  run<10>("v_lshrrev_b64", d_o, d_in, d_clk);
  run<11>("v_and_b32", d_o, d_in, d_clk);
  run<12>("v_alignbit_b32", d_o, d_in, d_clk);
  run<13>("v_add3_u32", d_o, d_in, d_clk);
  return 0;
}
