// Dev probe: fp32 atomic-add throughput by scope; per-XCD private copies (HW_REG_XCC_ID) with workgroup-scope atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7; }

template <int SCOPE, bool PRIV>
__global__ __launch_bounds__(256) void scatter(float* table, uint32_t mask, size_t copy_stride, int per_thread, int* xcc_seen) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  float* base = table;
  if (PRIV) {
    const int x = xcc_id();
    base += (size_t)x * copy_stride;
    if (threadIdx.x == 0) xcc_seen[blockIdx.x] = x;
  }
  for (int i = 0; i < per_thread; ++i) {
    const uint32_t idx = hash32(t * 131u + i) & mask;
    __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, SCOPE);
  }
}
__global__ void reduce8(const float* t, size_t n, size_t stride, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0;
  for (int c = 0; c < 8; ++c) s += t[c * stride + i];
  out[i] = s;
}
// GROUP consecutive lanes add to GROUP consecutive floats (same 64-byte line when GROUP <= 16)
template <int GROUP>
__global__ __launch_bounds__(256) void scatter_grouped(float* table, uint32_t mask, int per_thread) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    const uint32_t idx = ((hash32((t / GROUP) * 131u + i) * GROUP) + (t % GROUP)) & mask;
    __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <class F>
static float timeit(F&& f, int reps = 3) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  for (int logn : {12, 16, 20, 24}) {
    const size_t n = 1ull << logn;
    const uint32_t mask = (uint32_t)n - 1;
    const int G = 8192, PT = 64;
    const double ops = (double)G * 256 * PT;
    float *single, *priv, *out; int* seen;
    hipMalloc(&single, n * 4); hipMalloc(&priv, n * 4 * 8); hipMalloc(&out, n * 4); hipMalloc(&seen, G * 4);
    hipMemset(single, 0, n * 4); hipMemset(priv, 0, n * 32);
    float a = timeit([&] { scatter<__HIP_MEMORY_SCOPE_AGENT, false><<<G, 256>>>(single, mask, 0, PT, seen); });
    float b = timeit([&] { scatter<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<G, 256>>>(priv, mask, n, PT, seen); });
    float c = timeit([&] { scatter<__HIP_MEMORY_SCOPE_WAVEFRONT, true><<<G, 256>>>(priv, mask, n, PT, seen); });
    float d = timeit([&] { scatter<__HIP_MEMORY_SCOPE_AGENT, true><<<G, 256>>>(priv, mask, n, PT, seen); });
    {
      float g1 = timeit([&] { scatter_grouped<1><<<G, 256>>>(single, mask, PT); });
      float g2 = timeit([&] { scatter_grouped<2><<<G, 256>>>(single, mask, PT); });
      float g4 = timeit([&] { scatter_grouped<4><<<G, 256>>>(single, mask, PT); });
      float g16 = timeit([&] { scatter_grouped<16><<<G, 256>>>(single, mask, PT); });
      float g64 = timeit([&] { scatter_grouped<64><<<G, 256>>>(single, mask, PT); });
      printf("table 2^%d grouped lanes: 1: %.1f G/s  2: %.1f  4: %.1f  16: %.1f  64: %.1f G/s\n", logn, ops / g1 / 1e6, ops / g2 / 1e6, ops / g4 / 1e6, ops / g16 / 1e6, ops / g64 / 1e6);
    }
    // correctness of the private copies: zero, one launch of the workgroup-scope variant, reduce, compare with the single table
    hipMemset(single, 0, n * 4); hipMemset(priv, 0, n * 32);
    scatter<__HIP_MEMORY_SCOPE_AGENT, false><<<G, 256>>>(single, mask, 0, PT, seen);
    scatter<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<G, 256>>>(priv, mask, n, PT, seen);
    reduce8<<<(unsigned)((n + 255) / 256), 256>>>(priv, n, n, out);
    std::vector<float> h1(n), h2(n); std::vector<int> hs(G);
    hipMemcpy(h1.data(), single, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), out, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), seen, G * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; double tot = 0;
    for (size_t i = 0; i < n; ++i) { bad += h1[i] != h2[i]; tot += h2[i]; }
    int hist[8] = {0}; for (int x : hs) hist[x & 7]++;
    printf("table 2^%d: agent/single %.3f ms (%.1f G/s) | workgroup/private %.3f ms (%.1f G/s) | wavefront/private %.3f ms (%.1f G/s) | agent/private %.3f ms (%.1f G/s) | mismatches %zu total %.0f (expect %.0f) xcc hist %d %d %d %d %d %d %d %d\n",
           logn, a, ops / a / 1e6, b, ops / b / 1e6, c, ops / c / 1e6, d, ops / d / 1e6, bad, tot, ops, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    hipFree(single); hipFree(priv); hipFree(out); hipFree(seen);
  }
  return 0;
}
