// Instruction-fetch probe: straight-line code of a chosen size executed once per pass by every wave of a workgroup, against the
// same work as a short loop.  The fused geometry kernels are 330 - 380 KB of straight-line code (8 layers fully unrolled); the
// instruction cache is 64 KB per CU pair.  On most boxes the sequential prefetcher hides the misses, on the "slow" class
//  every kernel whose code exceeds the cache runs ~2.4x slower while small kernels are unaffected.  This
// probe measures cycles per instruction as a function of straight-line code size, with and without MFMA / LDS traffic around.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/probe_icache tools/probe_icache.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int N>
struct Rep {
  template <class F>
  static __device__ __forceinline__ void run(F&& f) {
    if constexpr (N >= 8) {
      Rep<N / 2>::run(f);
      Rep<N - N / 2>::run(f);
    } else {
      if constexpr (N >= 1) f();
      if constexpr (N >= 2) Rep<N - 1>::run(f);
    }
  }
};

// One "group" = 8 independent v_fma_f32 (VOP3, 8 bytes each) = 64 B of code = one cache line.  KB kilobytes of code = 16 KB groups.
template <int KB, bool LOOP>
__global__ __launch_bounds__(256, 1) void code_kernel(float* out, unsigned long long* clk, int passes, float s) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    if constexpr (LOOP) {
#pragma unroll 1
      for (int g = 0; g < KB * 16; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(s));
      }
    } else {
      Rep<KB * 16>::run([&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(s));
      });
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x < 1024) clk[blockIdx.x] = c1 - c0;
}

template <int KB, bool LOOP>
static void run(float* d_out, unsigned long long* d_clk, int wgs_per_cu) {
  const int passes = 2048 / KB > 2 ? 2048 / KB : 2;
  const unsigned grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  code_kernel<KB, LOOP><<<grid, 256>>>(d_out, d_clk, passes, 1e-9f);
  hipEventRecord(e0);
  code_kernel<KB, LOOP><<<grid, 256>>>(d_out, d_clk, passes, 1e-9f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(1024);
  hipMemcpy(c.data(), d_clk, 1024 * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < 1024; ++i) avg += (double)c[i];
  avg /= 1024;
  const double instr = (double)passes * KB * 128;  // per wave
  printf("%4d KB %-9s %d WG/CU: %8.3f ms  %6.2f cycles / instruction / wave  (%.0f cycles per 64 B line)\n", KB, LOOP ? "loop" : "straight",
         wgs_per_cu, ms, avg / instr, avg / instr * 8);
}

int main() {
  float* d_out;
  unsigned long long* d_clk;
  hipMalloc(&d_out, 256 * 4 * 256 * 4);
  hipMalloc(&d_clk, 1024 * 8);
  run<16, true>(d_out, d_clk, 1);
  run<16, false>(d_out, d_clk, 1);
  run<32, false>(d_out, d_clk, 1);
  run<48, false>(d_out, d_clk, 1);
  run<64, false>(d_out, d_clk, 1);
  run<96, false>(d_out, d_clk, 1);
  run<128, false>(d_out, d_clk, 1);
  run<256, false>(d_out, d_clk, 1);
  run<384, false>(d_out, d_clk, 1);
  run<384, false>(d_out, d_clk, 4);
  run<384, true>(d_out, d_clk, 1);
  return 0;
}
