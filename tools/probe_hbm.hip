// Dev probe: what HBM streaming rate can THIS box reach, and with which access shape?  (the plain copy of
// probe_box.hip reached 4.6 - 4.8 TB/s where MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy.)
// Variants: copy (16 B per lane) by grid size / loads in flight / non-temporal hints, read-only, write-only, and the access shape
// of the fused network kernels (a wave moves 4 KiB tile-packed blocks with 16 dword accesses of 256 B each) next to the same
// blocks moved with 4 dwordx4 accesses of 1 KiB.  Buffers are 2 GiB each (>> the 256 MiB Infinity Cache).  The last section
// repeats the best copy for ~3 s and prints the rate per ~0.3 s window (power / clock limiters show up as a decay).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy16(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
template <int U>
__global__ __launch_bounds__(256) void read16(const f32x4* __restrict__ src, float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = s[0];
}
__global__ __launch_bounds__(256) void write16(f32x4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}
// The fused kernels' shape: workgroup = 4 waves, each wave owns the tile blockIdx.x * 4 + wave and moves NR blocks in, NW blocks
// out (blocks of 4 KiB = [16][64] floats), block b of array a at (a * tiles + tile) * 1024 floats.  W4 = 0: 16 dword accesses
// per block (lane + 64 r), W4 = 1: 4 dwordx4 accesses per block (4 lane + 256 r).
template <bool W4>
__global__ __launch_bounds__(256, 1) void tp_stream(const float* __restrict__ src, float* __restrict__ dst, int nr, int nw, size_t tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t tile = (size_t)blockIdx.x * 4 + wave;
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int a = 0; a < nr; ++a) {
    const float* p = src + ((size_t)a * tiles + tile) * 1024;
    if constexpr (W4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * lane + 256 * r);
        acc[4 * r] += v[0]; acc[4 * r + 1] += v[1]; acc[4 * r + 2] += v[2]; acc[4 * r + 3] += v[3];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += p[lane + 64 * r];
    }
    if (a < nw) {
      float* q = dst + ((size_t)a * tiles + tile) * 1024;
      if constexpr (W4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(q + 4 * lane + 256 * r) = f32x4{acc[4 * r], acc[4 * r + 1], acc[4 * r + 2], acc[4 * r + 3]};
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) q[lane + 64 * r] = acc[r];
      }
    }
  }
}

template <class F>
static float timeit(F&& f, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}
int main(int argc, char** argv) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clock %d MHz  mem clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.memoryClockRate / 1000);
  const size_t bytes = 2ull << 30, n = bytes / 16;
  f32x4 *a, *b; float* out;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 4096);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  const double GB2 = 2.0 * bytes / 1e9, GB1 = bytes / 1e9;
  for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
    float m1 = timeit([&] { copy16<1, false><<<g, 256>>>(a, b, n); });
    float m4 = timeit([&] { copy16<4, false><<<g, 256>>>(a, b, n); });
    float m8 = timeit([&] { copy16<8, false><<<g, 256>>>(a, b, n); });
    float t4 = timeit([&] { copy16<4, true><<<g, 256>>>(a, b, n); });
    printf("copy 2 GiB, grid %5d x 256: 1 load in flight %.2f TB/s | 4: %.2f | 8: %.2f | 4 + nt: %.2f   (read + write)\n", g, GB2 / m1, GB2 / m4,
           GB2 / m8, GB2 / t4);
  }
  {
    float r4 = timeit([&] { read16<4><<<256 * 16, 256>>>(a, out, n); });
    float r8 = timeit([&] { read16<8><<<256 * 16, 256>>>(a, out, n); });
    float w = timeit([&] { write16<<<256 * 16, 256>>>(b, n); });
    printf("read-only: 4 in flight %.2f TB/s | 8: %.2f   write-only: %.2f TB/s\n", GB1 / r4, GB1 / r8, GB1 / w);
  }
  {
    const int narr = 32; const size_t tiles = bytes / (narr * 4096);  // 32 arrays of 64 MiB = 16384 tiles = 524288 points
    for (int nw : {8, 16, 32}) {
      float m0 = timeit([&] { tp_stream<false><<<(unsigned)(tiles / 4), 256>>>((const float*)a, (float*)b, narr, nw, tiles); });
      float m1 = timeit([&] { tp_stream<true><<<(unsigned)(tiles / 4), 256>>>((const float*)a, (float*)b, narr, nw, tiles); });
      const double gb = (double)(narr + nw) * tiles * 4096 / 1e9;
      printf("tile-packed blocks, one wave per SIMD, %d read + %d written per tile: dword %.2f TB/s | dwordx4 %.2f TB/s\n", narr, nw, gb / m0, gb / m1);
    }
  }
  {
    printf("sustained copy (4 in flight, grid 4096), TB/s per ~0.3 s window:");
    for (int w = 0; w < 10; ++w) {
      float ms = timeit([&] { copy16<4, false><<<4096, 256>>>(a, b, n); }, 300);
      printf(" %.2f", GB2 / ms);
    }
    printf("\n");
  }
  return 0;
}
