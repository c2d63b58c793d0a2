// Dev probe: per-box characterisation (boxes in the pool differ): pure MFMA rates, L2 -> LDS DMA stream rate, LDS read rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 1) void mfma_bf16(float* out, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256, 1) void mfma_f32(float* out, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  float a = threadIdx.x * 0.001f, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// every WG streams the same `bytes` (L2 resident) into LDS, chunk by chunk, `reps` times
__global__ __launch_bounds__(256, 1) void dma_stream(const float* src, size_t floats, int reps, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t chunks = floats / 8192;  // 32 KiB chunks
  int buf = 0;
  for (int r = 0; r < reps; ++r)
    for (size_t c = 0; c < chunks; ++c) {
      const float* g = src + c * 8192;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int piece = i * 4 + wave;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(lds + buf * 8192 + piece * 256), 16, 0, 0);
      }
      buf = (buf + 1) & 3;
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
__global__ __launch_bounds__(256, 1) void lds_read(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 s = {0, 0, 0, 0};
  const unsigned base = (unsigned)(size_t)(lds) + lane * 16;  // LDS byte address (address space 3 pointers are 32-bit offsets)
  for (int it = 0; it < iters; ++it) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(base + ((it & 3) << 14) * 0), "n"(u * 1024));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// dependent chain of ds_read_b32 (each address comes from the previous read): LDS latency per read in shader cycles,
// alone (MFMA = 0) or with independent bf16 MFMAs issued between the reads (MFMA = 1)
template <int MFMA>
__global__ __launch_bounds__(256, 1) void lds_latency(unsigned long long* cyc, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* li = reinterpret_cast<unsigned*>(lds);
  for (int i = threadIdx.x; i < 16384; i += 256) li[i] = ((i + 64 * 7) & 16383) * 4;  // byte offset of the next element
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  unsigned addr = (unsigned)(size_t)(lds) + threadIdx.x * 4;
  const unsigned base = (unsigned)(size_t)(lds);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; ++it) {
    unsigned nxt;
    asm volatile("ds_read_b32 %0, %1" : "=v"(nxt) : "v"(addr));
    if constexpr (MFMA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    addr = base + nxt;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x < 64) cyc[blockIdx.x] = t1 - t0;
  float s = addr;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// Access pattern of the fused network kernels: every wave reads one 4 KiB block from each of NARR arrays (same tile index) and
// writes one - ARRAY-MAJOR (NARR separate multi-GiB arrays: NARR distinct pages per wave) versus TILE-MAJOR (the NARR blocks of a
// tile contiguous: one or two pages per wave).  Same bytes; a gap between the two is address-translation (TLB) cost.
template <bool TILE_MAJOR>
__global__ __launch_bounds__(256, 1) void block_stream(const float* __restrict__ src, float* __restrict__ dst, int narr, size_t tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t tile = (size_t)blockIdx.x * 4 + wave;
  float acc = 0.f;
  for (int a = 0; a < narr; ++a) {
    const size_t blk = TILE_MAJOR ? tile * narr + a : (size_t)a * tiles + tile;
    const float* p = src + blk * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += p[r * 64];
    if ((a & 3) == 0) {
      float* q = dst + blk * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r * 64] = acc;
    }
  }
  if (acc == 123.456f) dst[0] = acc;
}
__global__ __launch_bounds__(256) void hbm_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
template <class F>
static float timeit(F&& f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clock %d MHz  mem clock %d MHz  bus %d  L2 %d KiB  total %.1f GiB  gcn %s\n", p.name, p.multiProcessorCount, p.clockRate / 1000,
         p.memoryClockRate / 1000, p.memoryBusWidth, p.l2CacheSize / 1024, p.totalGlobalMem / 1073741824.0, p.gcnArchName);
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  float* src; hipMalloc(&src, 16 << 20); hipMemset(src, 0, 16 << 20);
  const int G = 4096;
  {
    const int iters = 400;
    float ms = timeit([&] { mfma_bf16<<<G, 256>>>(out, iters); });
    printf("mfma bf16 32x32x16 : %7.3f ms  %7.1f TFLOP/s\n", ms, (double)G * 4 * iters * 32 * 32768.0 / ms / 1e9);
    ms = timeit([&] { mfma_f32<<<G, 256>>>(out, iters); });
    printf("mfma f32 32x32x2   : %7.3f ms  %7.1f TFLOP/s\n", ms, (double)G * 4 * iters * 32 * 4096.0 / ms / 1e9);
  }
  {
    float4 *a, *b; const size_t n = (1ull << 30) / 16;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    float ms = timeit([&] { hbm_copy<<<256 * 16, 256>>>(a, b, n); });
    printf("HBM copy 1 GiB     : %7.3f ms  %6.2f TB/s (read + write)\n", ms, 2.0 * n * 16 / ms / 1e9);
    hipFree(a); hipFree(b);
  }
  {
    // sustained matrix load: bf16 MFMA back to back for ~1.5 s, rate per ~150 ms window (power / current limiters show up here)
    printf("sustained bf16 mfma:");
    for (int w = 0; w < 10; ++w) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int i = 0; i < 50; ++i) mfma_bf16<<<G, 256>>>(out, 400);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf(" %.0f", (double)G * 4 * 400 * 32 * 32768.0 * 50 / ms / 1e9);
    }
    printf(" TFLOP/s\n");
  }
  hipFuncSetAttribute((const void*)dma_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (size_t mb : {1, 2, 4, 16}) {
    const int reps = 2;
    float ms = timeit([&] { dma_stream<<<G, 256, 131072>>>(src, (mb << 20) / 4, reps, out); });
    printf("L2->LDS DMA stream, %zu MiB working set: %7.3f ms  %6.2f TB/s\n", mb, ms, (double)G * reps * (mb << 20) / ms / 1e9);
  }
  {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)lds_read, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    float ms = timeit([&] { lds_read<<<G, 256, 65536>>>(out, iters); });
    const double bytes = (double)G * 256 * iters * 16 * 16;
    printf("LDS ds_read_b128   : %7.3f ms  %6.2f TB/s  = %.0f B/clk/CU at the nominal %d MHz (4 waves per CU, 16 reads per wait)\n", ms, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / p.multiProcessorCount / (p.clockRate * 1e3), p.clockRate / 1000);
  }
  {
    const int narr = 40; const size_t tiles = 16384;  // 524288 points: 40 x 64 MiB
    float *a, *b; hipMalloc(&a, narr * tiles * 4096); hipMalloc(&b, narr * tiles * 4096); hipMemset(a, 0, narr * tiles * 4096);
    float m0 = timeit([&] { block_stream<false><<<(unsigned)(tiles / 4), 256>>>(a, b, narr, tiles); });
    float m1 = timeit([&] { block_stream<true><<<(unsigned)(tiles / 4), 256>>>(a, b, narr, tiles); });
    const double bytes = 1.25 * narr * tiles * 4096.0;
    printf("4 KiB blocks of 40 arrays per wave: array-major %.3f ms (%.2f TB/s), tile-major %.3f ms (%.2f TB/s)\n", m0, bytes / m0 / 1e9, m1,
           bytes / m1 / 1e9);
    hipFree(a); hipFree(b);
  }
  {
    unsigned long long* cyc; hipMalloc(&cyc, 64 * 8);
    hipFuncSetAttribute((const void*)lds_latency<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)lds_latency<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<unsigned long long> h(64);
    lds_latency<0><<<256, 256, 65536>>>(cyc, out); hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
    double a0 = 0; for (auto v : h) a0 += (double)v; a0 /= 64 * 512;
    lds_latency<1><<<256, 256, 65536>>>(cyc, out); hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
    double a1 = 0; for (auto v : h) a1 += (double)v; a1 /= 64 * 512;
    printf("LDS dependent ds_read_b32 : %.0f cycles per read alone, %.0f with 4 bf16 MFMAs (128 cycles of matrix pipe) per read\n", a0, a1);
  }
  return 0;
}
